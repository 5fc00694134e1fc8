"""MIGD: the flat container the golden-dump tool and the tests exchange arrays in (no zip, no numpy in Rust).

    file   := "MIGD" u32(version = 1) u32(n_records) record*
    record := u32(name_len) name_bytes u32(dtype) u64(count) payload          (little endian, payload unpadded)
    dtype  := 0 u8 | 1 u32 | 2 f32 | 3 u64

tools/golden_dump/src/main.rs reads and writes the same layout."""
import struct

import numpy as np

DTYPES = {0: np.uint8, 1: np.uint32, 2: np.float32, 3: np.uint64}
CODES = {np.dtype(v): k for k, v in DTYPES.items()}


def write(path, arrays):
    with open(path, "wb") as f:
        f.write(b"MIGD" + struct.pack("<II", 1, len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<IQ", CODES[a.dtype], a.size) + a.tobytes())


def read(path):
    buf = open(path, "rb").read()
    assert buf[:4] == b"MIGD", "not a MIGD file"
    version, n = struct.unpack_from("<II", buf, 4)
    assert version == 1
    off, out = 12, {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", buf, off)
        name = buf[off + 4:off + 4 + ln].decode()
        code, count = struct.unpack_from("<IQ", buf, off + 4 + ln)
        off += 4 + ln + 12
        dt = np.dtype(DTYPES[code])
        out[name] = np.frombuffer(buf, dt, count, off).copy()
        off += count * dt.itemsize
    assert off == len(buf), "trailing bytes"
    return out
