"""Shared pieces of the bench workloads: constants, the Workload record, byte models of the flat frame kernel."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
PROFILED_BLOCKS = 3
N_FRAMES = 256          # distinct prepared camera frames, cycled


def flat_bytes_per_entity(n_views, fused=True):
    # frame kernel: read Aabb 24 + flags 1 + layers 4 + vv 1 and T 40 (fused) or resident G 48 (unfused);
    # write vv 1 + (V view masks + vv change mask)/8 bits + V/64 wave counts, and G 48 + its change mask (fused)
    rd = 30.0 + (40.0 if fused else 48.0)
    wr = 1.0 + (n_views + 1) / 8.0 + n_views / 64.0 + ((48.0 + 1.0 / 8.0) if fused else 0.0)
    return rd + wr


# With the row summary (kernels.h RowSummary) a wave whose 64 rows agree in Aabb / flags / RenderLayers reads 32 bytes instead of
# 64 x (24 + 1 + 4): what the kernel moves for such rows is 28.5 B less than the algorithmic figure, which stays SURVEY 8(d)'s.
ROW_SUMMARY_SAVES = 29.0 - 32.0 / 64.0



class Workload:
    """step(f) enqueues one frame; units = work items per frame on this rank; rows = rows the dominant kernel streams."""

    def __init__(self, name, step, units, bytes_per_row, dominant, config, metric, unit, rows=None, kernels=None):
        self.name, self.step, self.units, self.bytes_per_row = name, step, units, bytes_per_row
        self.dominant, self.config, self.metric, self.unit = dominant, config, metric, unit
        self.rows = units if rows is None else rows
        self.kernels = kernels or [dominant]   # what the profiled blocks time


def camera_frusta(n_views, frame):
    from bevy_amd import api, workloads as W
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([api.compute_frustum(cfv, W.many_cubes_camera(frame, yaw=v * np.pi / 2), W.CAMERA_FAR) for v in range(n_views)])




def with_args(args, **kw):
    import copy
    a = copy.copy(args)
    for k, v in kw.items():
        setattr(a, k, v)
    return a
