"""Entity-range sharding over the GPUs of one node: one process per GPU, rows split into
contiguous 256-aligned ranges, every GPU culls its own rows against ALL views, and ONE all-gather (RCCL over
xGMI; gloo in the CPU tests) of the packed ViewVisibility bitmask gives every rank the full per-view masks.
Payload is tiny (10M rows x 4 views = 5 MB in total) so the exchange is latency-bound: a single
all_gather_into_tensor on an in-place layout, no bucketing.

Layout of the gathered buffer (uint64 words): [world][n_views][words_per_shard]; rank r's kernels write
straight into block r (mi_bind_visibility_output), so the collective is in place and needs no packing pass.
"""
import os

import numpy as np

ROW_ALIGN = 256  # one workgroup; keeps every shard's first row on a 64-bit mask word boundary


def shard_rows(n_rows, world, rank):
    """Contiguous row range [lo, hi) of `rank`: ceil-divided, aligned to ROW_ALIGN (the last shards may be short
    or empty)."""
    per = -(-n_rows // world)
    per = -(-per // ROW_ALIGN) * ROW_ALIGN
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi


def words_per_shard(n_rows, world):
    lo, hi = shard_rows(n_rows, world, 0)
    per = -(-max(hi - lo, 1) // ROW_ALIGN) * ROW_ALIGN
    return per // 64


def gathered_words(n_rows, world, n_views):
    return world * n_views * words_per_shard(n_rows, world)


def block_offset_words(n_rows, world, n_views, rank):
    """(words_per_view, word_offset) to pass to mi_bind_visibility_output for this rank."""
    w = words_per_shard(n_rows, world)
    return w, rank * n_views * w


def all_gather_visibility(full, n_rows, world, n_views, rank, group=None):
    """In-place all-gather of the [world][n_views][W] uint64 buffer `full` (a torch tensor viewed as int64)."""
    import torch.distributed as dist
    w = words_per_shard(n_rows, world)
    blk = n_views * w
    if world == 1:
        return full
    dist.all_gather_into_tensor(full, full[rank * blk:(rank + 1) * blk], group=group)
    return full


def _dist_on():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:  # noqa: BLE001
        return False


class MaskGatherer:
    """The per-frame exchange, pipelined: frame f's all-gather runs on its own HIP stream while the kernels of the
    following frames run on the compute stream (n_bufs gathered buffers rotate), so a step costs
    max(kernels, collective), not their sum.  The collective is ONE in-place all-gather per frame.

    On GPUs it is issued by calling RCCL directly (ncclAllGather through ctypes on the library torch already
    loaded): the masks are ~125 KB per GPU, so the exchange is latency-bound and torch.distributed's Python
    dispatch (tens of microseconds per call) would otherwise be the longest thing in the frame.  The communicator
    is bootstrapped with torch.distributed (the ncclUniqueId is broadcast from rank 0) and checked once against a
    known pattern; if anything in that path fails the gatherer falls back to dist.all_gather_into_tensor.
    On CPU tensors (gloo tests) it always uses torch.distributed.
    """

    def __init__(self, n_rows, world, n_views, rank, device=None, direct=True, group=None, n_bufs=4, n_comms=None, pipelined=None):
        import torch
        self.torch = torch
        self.world, self.rank, self.n_views, self.n_rows, self.group = world, rank, n_views, n_rows, group
        self.w = words_per_shard(n_rows, world)
        self.block = n_views * self.w                       # int64 words per rank
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        dev = device if device is not None else "cpu"
        self.n_bufs = n_bufs
        self.bufs = [torch.zeros(world * self.block, dtype=torch.int64, device=dev) for _ in range(n_bufs)]
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        self.ev_kernels = [torch.cuda.Event() for _ in range(n_bufs)] if self.on_gpu else None
        self.ev_gathered = [torch.cuda.Event() for _ in range(n_bufs)] if self.on_gpu else None
        self.rccl = None
        self.comms = []
        # How the library drives the exchange when it is attached to a context (mi_exchange_set_mode): the SIMPLE mode -- one
        # communicator, one stream, event-ordered -- unless pipelined=True / MI_XCH_MODE=pipelined asks for the round-1
        # machinery (exchange thread, alternating communicators (MI_XCH_COMMS=1..4, default 2), kernel-signalled completion),
        # which has only ever been measured against a 1-rank communicator.
        self.pipelined = (os.environ.get("MI_XCH_MODE", "simple") == "pipelined") if pipelined is None else bool(pipelined)
        default_comms = "1"  # (two alternating communicators: MI_XCH_COMMS=2 -- no faster on one GPU, and one ordering hazard more on eight)
        self.n_comms = int(os.environ.get("MI_XCH_COMMS", default_comms)) if n_comms is None else n_comms
        self.n_comms = max(1, min(4, self.n_comms))
        self.native = False
        self.mode = "torch.distributed"
        self.fallback_reason = None
        if self.on_gpu and direct and os.environ.get("MI_DIRECT_RCCL", "1") != "0":
            ok = 1
            try:
                self._init_rccl()
                self._self_check()
            except Exception as e:  # noqa: BLE001 -- any failure here means "use the portable path"
                ok = 0
                self.fallback_reason = repr(e)
            if _dist_on():  # every rank must take the same path
                import torch.distributed as dist
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if ok:
                self.mode = "rccl-direct"
            else:
                self.rccl = None
                self.fallback_reason = self.fallback_reason or "another rank failed to set up direct RCCL"

    def attach(self, ctx, pipelined=None):
        """Hands the per-frame exchange to the library (mi_exchange_configure): one FFI call per frame then does
        bind + kernels + events + ncclAllGather natively.  Only in rccl-direct mode; returns whether it did."""
        if self.rccl is None:
            return False
        import ctypes as C
        if pipelined is not None:
            self.pipelined = bool(pipelined)
        fn = C.cast(self.rccl.ncclAllGather, C.c_void_p).value
        if self.native:  # re-attaching in another mode: off first (drains what is in flight)
            ctx.exchange_configure(None, None, None, 0, 0, 0, 0)
        ctx.exchange_set_mode(1 if self.pipelined else 0)
        comms = self.comms if self.pipelined else self.comms[:1]
        ctx.exchange_configure([c.value for c in comms], fn, [b.data_ptr() for b in self.bufs], self.w, self.rank * self.block,
                               self.block * 8, self.rank)
        self.mode = "rccl-native-pipelined" if self.pipelined else "rccl-native"
        self.native = True
        return True

    def calibrate(self, ctx, run_frames, frames=40, margin=0.93):
        """The rule that picks the exchange mode, applied by measurement at start-up (every rank runs the same frames in lockstep, so
        the collectives match): `frames` frames through the SIMPLE mode (events between the streams: two marker packets in the
        compute queue per frame, ~10 us of device time whatever the kernel) and through the PIPELINED mode on ONE communicator (no
        packet in the compute queue: the next frame's launch publishes "masks complete", the communication stream waits on the
        value) -- pipelined where it is faster by more than 1 - margin ON EVERY RANK (a MIN-reduce of the verdicts: all ranks must
        take the same path), simple otherwise.  run_frames(k) must enqueue k frames and return after they have completed.
        MI_XCH_MODE=simple|pipelined in the environment skips the measurement.  Returns the record bench.py prints."""
        import time
        forced = os.environ.get("MI_XCH_MODE")
        if self.rccl is None or not self.native or forced in ("simple", "pipelined"):
            return {"chosen": "pipelined" if self.pipelined else "simple", "rule": "MI_XCH_MODE" if forced else "no direct RCCL: nothing to choose"}
        timings = {}
        for name, pipe in (("simple", False), ("pipelined", True)):
            self.attach(ctx, pipelined=pipe)
            run_frames(10)
            t0 = time.perf_counter()
            run_frames(frames)
            timings[name] = (time.perf_counter() - t0) / frames
        want = 1 if timings["pipelined"] < margin * timings["simple"] else 0
        if _dist_on():
            import torch.distributed as dist
            flag = self.torch.tensor([want], dtype=self.torch.int32, device=self.bufs[0].device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            want = int(flag.item())
        self.attach(ctx, pipelined=bool(want))
        return {"chosen": "pipelined" if want else "simple", "simple_us_per_frame": round(1e6 * timings["simple"], 2),
                "pipelined_us_per_frame": round(1e6 * timings["pipelined"], 2),
                "rule": f"pipelined (one communicator, value-signalled) iff faster than {margin} x simple on every rank, {frames} frames each at start-up"}

    def rccl_ranks(self):
        """Ranks of the communicator as RCCL reports them (ncclCommCount); the torch.distributed world size on the portable path."""
        if self.rccl is None:
            return self.world
        import ctypes as C
        n = C.c_int(0)
        try:
            self.rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            return int(n.value) if self.rccl.ncclCommCount(self.comm, C.byref(n)) == 0 else None
        except AttributeError:
            return None

    # -- layout -------------------------------------------------------------------------------------
    def bind_args(self, frame):
        """(device_ptr, words_per_view, word_offset) for mi_bind_visibility_output of this frame's buffer."""
        return self.bufs[frame % self.n_bufs].data_ptr(), self.w, self.rank * self.block

    def buffer(self, frame):
        return self.bufs[frame % self.n_bufs]

    # -- direct RCCL ----------------------------------------------------------------------------------
    def _init_rccl(self):
        import ctypes as C
        torch = self.torch
        import torch.distributed as dist
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(path if os.path.exists(path) else "librccl.so")

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        comms = []
        for _ in range(self.n_comms):  # every rank creates them in the same order
            uid = UniqueId()
            t = torch.zeros(129, dtype=torch.uint8, device=self.bufs[0].device)  # 128 id bytes + "rank 0 has an id"
            if self.rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) == 0:
                t[:128].copy_(torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8))
                t[128] = 1
            if _dist_on():
                dist.broadcast(t, 0, group=self.group)  # always reached by every rank
            host = t.cpu().numpy()
            if host[128] != 1:
                raise RuntimeError("ncclGetUniqueId failed on rank 0")
            C.memmove(C.byref(uid), host[:128].tobytes(), 128)
            comm = C.c_void_p()
            rc = lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank)
            if rc:
                raise RuntimeError(f"ncclCommInitRank: {lib.ncclGetErrorString(rc)}")
            comms.append(comm)
        self.comms = comms
        comm = comms[0]
        self.rccl, self.comm = lib, comm

    def _rccl_all_gather(self, buf, stream_handle, comm=None):
        nbytes = self.block * 8
        base = buf.data_ptr()
        rc = self.rccl.ncclAllGather(base + self.rank * nbytes, base, nbytes, 1, comm or self.comm, stream_handle)  # 1 = ncclUint8
        if rc:
            raise RuntimeError(f"ncclAllGather: {self.rccl.ncclGetErrorString(rc)}")

    def _self_check(self):
        torch = self.torch
        buf = self.bufs[0]
        for k, comm in enumerate(self.comms):  # every communicator, in order
            buf.zero_()
            buf[self.rank * self.block:(self.rank + 1) * self.block] = self.rank + 1 + 100 * k
            torch.cuda.current_stream().synchronize()
            self._rccl_all_gather(buf, self.comm_stream.cuda_stream, comm)
            self.comm_stream.synchronize()
            expect = (torch.arange(1, self.world + 1, dtype=torch.int64, device=buf.device) + 100 * k).repeat_interleave(self.block)
            if not torch.equal(buf, expect):
                raise RuntimeError(f"direct RCCL all-gather self-check mismatch (communicator {k})")
        buf.zero_()
        torch.cuda.current_stream().synchronize()

    # -- per frame ------------------------------------------------------------------------------------
    def before_kernels(self, frame, compute_stream=None):
        """The buffer of this frame was last used two frames ago: its all-gather must have drained."""
        if self.on_gpu:
            (compute_stream or self.torch.cuda.current_stream()).wait_event(self.ev_gathered[frame % self.n_bufs])

    def after_kernels(self, frame, compute_stream=None):
        """Enqueue this frame's all-gather behind its kernels, on the communication stream."""
        i = frame % self.n_bufs
        buf = self.bufs[i]
        if not self.on_gpu:
            if self.world > 1:
                import torch.distributed as dist
                dist.all_gather_into_tensor(buf, buf[self.rank * self.block:(self.rank + 1) * self.block].clone(), group=self.group)
            return buf
        cs = compute_stream or self.torch.cuda.current_stream()
        self.ev_kernels[i].record(cs)
        self.comm_stream.wait_event(self.ev_kernels[i])
        if self.rccl is not None:
            self._rccl_all_gather(buf, self.comm_stream.cuda_stream)
        elif self.world > 1:
            import torch.distributed as dist
            with self.torch.cuda.stream(self.comm_stream):
                dist.all_gather_into_tensor(buf, buf[self.rank * self.block:(self.rank + 1) * self.block], group=self.group)
        self.ev_gathered[i].record(self.comm_stream)
        return buf

    def synchronize(self):
        if self.on_gpu:
            self.comm_stream.synchronize()

    def all_gather_latency_us(self, reps=50):
        """The collective by itself: `reps` all-gathers of one frame's masks back to back on the communication stream, between two
        events on THAT stream (the stream the all-gather runs on) -- what the exchange costs when nothing hides it.  Every rank must
        call it (it is a collective); a scratch buffer of the gathered size, so no frame's masks are touched.  None without direct
        RCCL."""
        if self.rccl is None or not self.on_gpu:
            return None
        torch = self.torch
        scratch = torch.zeros_like(self.bufs[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            self._rccl_all_gather(scratch, self.comm_stream.cuda_stream)
        self.comm_stream.synchronize()
        e0.record(self.comm_stream)
        for _ in range(reps):
            self._rccl_all_gather(scratch, self.comm_stream.cuda_stream)
        e1.record(self.comm_stream)
        self.comm_stream.synchronize()
        return round(1e3 * e0.elapsed_time(e1) / reps, 2)

    def unpack_gathered(self, words):
        """[world][n_views][w] uint64 words (host) -> bool [n_views][n_rows]: the masks of the whole scene as the gather holds them."""
        w = np.asarray(words).view(np.uint64).reshape(self.world, self.n_views, self.w)
        out = np.zeros((self.n_views, self.n_rows), bool)
        for r in range(self.world):
            lo, hi = shard_rows(self.n_rows, self.world, r)
            if hi > lo:
                bits = np.unpackbits(w[r].view(np.uint8), axis=1, bitorder="little")[:, :hi - lo]
                out[:, lo:hi] = bits.astype(bool)
        return out

    def close(self):
        if self.rccl is not None:
            self.synchronize()
            for c in self.comms:
                self.rccl.ncclCommDestroy(c)
            self.comms = []
            self.rccl = None


def unpack_view(full_words, n_rows, world, n_views, view):
    """numpy helper: the gathered buffer -> uint8[n_rows] visibility of one view (rows in global order)."""
    w = words_per_shard(n_rows, world)
    full_words = np.asarray(full_words).view(np.uint64).reshape(world, n_views, w)
    out = np.zeros(n_rows, np.uint8)
    for r in range(world):
        lo, hi = shard_rows(n_rows, world, r)
        if hi <= lo:
            continue
        bits = np.unpackbits(full_words[r, view].view(np.uint8), bitorder="little")
        out[lo:hi] = bits[:hi - lo]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Hierarchies: shard by root subtree.  Forest roots are independent
# (propagate_parent_transforms, systems.rs:522), so whole trees go to GPUs; a tree bigger than a GPU's fair share
# is opened up: its root is replicated (every rank that owns something below it recomputes it -- a handful of
# rows) and its child subtrees are distributed instead.  No collective: a rank's rows never read another rank's.
# ---------------------------------------------------------------------------------------------------------------------
NO_PARENT = 0xFFFFFFFF


def shard_hierarchy(parent, level_offsets, world, rank=None, slack=1.10):
    """parent[i] (NO_PARENT for roots) with rows in level (BFS) order, level_offsets as mi_upload_hierarchy takes them.
    Returns for `rank` (or a list for every rank if rank is None) a dict:
      rows           global row ids this rank holds, ascending (level order is kept, parents precede children)
      owned          bool per local row: this rank is responsible for the row's GlobalTransform (every global row is
                     owned by exactly one rank; replicated ancestors are owned by the lowest rank holding them)
      parent         local parent indices (NO_PARENT for roots)
      level_offsets  local level offsets
    Greedy: units start as the forest's trees and are placed biggest first on the least loaded rank; while the fullest
    rank exceeds slack x (nodes / world), the biggest unit that has children is replaced by its child subtrees (its
    root becomes a replicated row) and the placement is redone."""
    parent = np.asarray(parent, np.uint32)
    lv = np.asarray(level_offsets, np.int64)
    n = len(parent)
    is_root = parent == NO_PARENT
    # subtree sizes, deepest level first
    size = np.ones(n, np.int64)
    for l in range(len(lv) - 2, 0, -1):
        lo, hi = lv[l], lv[l + 1]
        np.add.at(size, parent[lo:hi].astype(np.int64), size[lo:hi])
    children_start = None  # children of a node are found by scanning its level's successor: build an index once
    order = np.argsort(np.where(is_root, -1, parent.astype(np.int64)), kind="stable")
    sorted_parent = np.where(is_root, -1, parent.astype(np.int64))[order]
    first = np.searchsorted(sorted_parent, np.arange(n), "left")
    last = np.searchsorted(sorted_parent, np.arange(n), "right")

    def children(i):
        return order[first[i]:last[i]]

    # units: (size, root).  Pack biggest-first on the least loaded rank; while the fullest rank exceeds
    # slack x (nodes / world), open the biggest unit that still has children and pack again.
    units = [(int(size[i]), int(i)) for i in np.nonzero(is_root)[0]]
    replicated = []
    ideal = n / max(world, 1)

    def pack(us):
        load = [0] * world
        where = {}
        for sz, i in sorted(us, key=lambda u: (-u[0], u[1])):
            r = min(range(world), key=lambda k: (load[k], k))
            load[r] += sz
            where[i] = r
        return load, where

    load, unit_rank = pack(units)
    for _ in range(64 * max(world, 1)):
        if world == 1 or max(load) <= slack * ideal:
            break
        splittable = [u for u in units if last[u[1]] > first[u[1]]]
        if not splittable:
            break
        big = max(splittable, key=lambda u: (u[0], -u[1]))
        units.remove(big)
        replicated.append(big[1])
        units.extend((int(size[c]), int(c)) for c in children(big[1]))
        load, unit_rank = pack(units)
    # rank of every node: the unit it lies in (replicated nodes: -1), level by level
    node_rank = np.full(n, -2, np.int64)
    rep = np.zeros(n, bool)
    rep[replicated] = True
    unit_root_rank = np.full(n, -2, np.int64)
    for i, r in unit_rank.items():
        unit_root_rank[i] = r
    for l in range(len(lv) - 1):
        lo, hi = lv[l], lv[l + 1]
        idx = np.arange(lo, hi)
        inherit = np.where(is_root[lo:hi], -2, node_rank[np.where(is_root[lo:hi], 0, parent[lo:hi]).astype(np.int64)])
        node_rank[lo:hi] = np.where(rep[idx], -1, np.where(unit_root_rank[idx] >= 0, unit_root_rank[idx], inherit))
    assert np.all(node_rank >= -1), "every row is in a unit or replicated"
    # which ranks need each replicated node: those owning something below it
    need = np.zeros((world, n), bool)
    for r in range(world):
        need[r] = node_rank == r
    for l in range(len(lv) - 2, 0, -1):
        lo, hi = lv[l], lv[l + 1]
        p = parent[lo:hi].astype(np.int64)
        for r in range(world):
            np.logical_or.at(need[r], p, need[r, lo:hi])
    out = []
    rep_owner = np.full(n, -1, np.int64)
    for i in replicated:
        holders = [r for r in range(world) if need[r, i]]
        rep_owner[i] = holders[0] if holders else 0
        if not holders:
            need[0, i] = True
    for r in range(world):
        rows = np.nonzero(need[r])[0].astype(np.uint32)
        local_of = np.full(n, NO_PARENT, np.uint32)
        local_of[rows] = np.arange(len(rows), dtype=np.uint32)
        p = parent[rows]
        lp = np.where(p == NO_PARENT, NO_PARENT, local_of[np.where(p == NO_PARENT, 0, p)]).astype(np.uint32)
        assert not np.any((p != NO_PARENT) & (lp == NO_PARENT)), "a held row's parent must be held"
        lvl = np.searchsorted(rows, lv.astype(np.uint32), "left").astype(np.uint32)
        lvl = np.unique(lvl) if len(rows) else np.zeros(1, np.uint32)  # drop empty levels
        if lvl[0] != 0:
            lvl = np.concatenate([[0], lvl]).astype(np.uint32)
        owned = (node_rank[rows] == r) | (rep_owner[rows] == r)
        out.append(dict(rows=rows, owned=owned, parent=lp, level_offsets=lvl.astype(np.uint32)))
    return out if rank is None else out[rank]


# ---------------------------------------------------------------------------------------------------------------------
# Light-cluster assignment sharded over the clusterable objects
# ---------------------------------------------------------------------------------------------------------------------
def shard_objects(n_objects, world, rank):
    """Contiguous range [lo, hi) of the gathered object list (assign.rs:190-296: point lights, spot lights, rect lights, probes,
    decals, in that order) that `rank` assigns.  Contiguous ranges keep the reference's push order: a cluster's entities are
    pushed in list order (assign.rs:740-800), i.e. rank 0's objects of the cluster first, then rank 1's, each in local order."""
    per = -(-n_objects // world) if world else n_objects
    lo = min(n_objects, rank * per)
    return lo, min(n_objects, lo + per)


def merge_cluster_assignments(local, first_object, world, rank, group=None, device=None, always_exchange=False):
    """Every rank has assigned ITS objects (`local` = what mi_cluster_assign / mi_cluster_download return for them: offsets[C + 1],
    indices[total] as LOCAL object numbers, counts[C, 6], farthest_z, total); this builds the assignment of the whole list on every
    rank with three collectives (RCCL over xGMI on GPUs -- device = the rank's cuda device --, gloo in the CPU tests):

      1. all-gather of the ranks' per-cluster entry counts ([C] words each: 13.8 KB at 16 x 9 x 24) and, with them, of the six
         per-type counts (summed), farthest_z (max) and the totals;
      2. (local) global offsets = prefix over the clusters of the summed counts; rank r's entries of cluster c start at
         offsets[c] + the counts of ranks < r in that cluster -- order within a cluster = (rank, local order) = the unsharded push
         order, because the shards are contiguous ranges of the gathered list;
      3. all-gather of the ranks' index segments (padded to the longest), scattered into place with the object numbers made global.

    Returns (offsets[C + 1] uint32, indices[total] uint32, counts[C, 6] uint32, farthest_z float32, total int): identical to the
    unsharded mi_cluster_assign of the concatenated list.  The reference has no counterpart (one process)."""
    import torch
    import torch.distributed as dist
    off_l, idx_l, cnt_l, far_l, tot_l = local
    off_l = np.asarray(off_l, np.int64)
    n_clusters = len(off_l) - 1
    per_cluster = torch.from_numpy(np.diff(off_l)).to(device)
    counts6 = torch.from_numpy(np.ascontiguousarray(np.asarray(cnt_l, np.int64).reshape(n_clusters, 6))).to(device)
    far = torch.tensor([float(far_l)], dtype=torch.float32, device=device)
    head = torch.tensor([int(tot_l), int(first_object)], dtype=torch.int64, device=device)
    if world == 1 and not always_exchange:  # (always_exchange: run the collectives on a one-rank group all the same -- tests)
        idx = np.asarray(idx_l[:int(tot_l)], np.uint32) + np.uint32(first_object)
        return off_l.astype(np.uint32), idx, np.asarray(cnt_l, np.uint32).reshape(n_clusters, 6), np.float32(far_l), int(tot_l)
    # ---- 1. counts ----
    all_counts = torch.empty(world * n_clusters, dtype=torch.int64, device=device)  # (flat: what every backend's all-gather takes)
    dist.all_gather_into_tensor(all_counts, per_cluster, group=group)
    all_counts = all_counts.view(world, n_clusters)
    heads = torch.empty(world * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(heads, head, group=group)
    heads = heads.view(world, 2)
    dist.all_reduce(counts6, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(far, op=dist.ReduceOp.MAX, group=group)
    # ---- 2. where everybody's entries go ----
    cluster_tot = all_counts.sum(0)
    offsets = torch.zeros(n_clusters + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(cluster_tot, 0)
    rank_before = torch.cumsum(all_counts, 0) - all_counts  # [world, C]: entries of lower ranks in each cluster
    totals = heads[:, 0]
    total = int(totals.sum().item())
    longest = int(totals.max().item())
    # ---- 3. segments ----
    seg = torch.zeros(max(longest, 1), dtype=torch.int64, device=device)
    seg[:int(tot_l)] = torch.from_numpy(np.asarray(idx_l[:int(tot_l)], np.int64)).to(device)
    segs = torch.empty(world * max(longest, 1), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(segs, seg, group=group)
    segs = segs.view(world, max(longest, 1))
    out = torch.zeros(max(total, 1), dtype=torch.int64, device=device)
    for r in range(world):
        t_r = int(totals[r].item())
        if t_r == 0:
            continue
        c_r = all_counts[r]
        local_off = torch.cumsum(c_r, 0) - c_r  # rank r's own CSR offsets
        # entry j of rank r lies in the cluster its offsets say; its place = offsets[c] + rank_before[r, c] + (j - local_off[c])
        shift = torch.repeat_interleave(offsets[:-1] + rank_before[r] - local_off, c_r)
        out[shift + torch.arange(t_r, device=device)] = segs[r, :t_r] + heads[r, 1]
    return (offsets.cpu().numpy().astype(np.uint32), out[:total].cpu().numpy().astype(np.uint32), counts6.cpu().numpy().astype(np.uint32),
            np.float32(far.item()), total)


def cluster_assign_sharded(ctx, view, pos_range, obj_type=None, layer_mask=None, spot_dir=None, spot_sin_cos=None, world=1, rank=0, group=None,
                           device=None, always_exchange=False):
    """assign_objects_to_clusters with the gathered object list range-sharded over the ranks: this rank's context assigns objects
    [lo, hi) (mi_cluster_assign), merge_cluster_assignments builds the whole view's assignment on every rank.  All arrays are the FULL
    list (every rank gathers the same list from its copy of the World); only the slice is uploaded."""
    n = len(pos_range) // 4
    lo, hi = shard_objects(n, world, rank)

    def cut(a, k):
        return None if a is None else np.ascontiguousarray(np.asarray(a).reshape(n, k)[lo:hi]).reshape(-1)
    local = ctx.cluster_assign(view, cut(pos_range, 4), cut(obj_type, 1), cut(layer_mask, 1), cut(spot_dir, 3), cut(spot_sin_cos, 2))
    return merge_cluster_assignments(local, lo, world, rank, group, device, always_exchange)
