"""configs[2], cluster stage alone: assign_objects_to_clusters over an uploaded object list."""
from .common import Workload


def build_lights(ctx, args):
    from bevy_amd import api, workloads as W
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    lights = W.many_lights(args.lights, 50.0, 0.3)
    cam = W.many_cubes_camera(0)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    ctx.cluster_upload_objects(lights)
    ctx.cluster_upload_view(view)

    def step(f):
        ctx.cluster_assign_resident()
    config = {"workload": f"many_lights-shaped: {args.lights} point lights (range 0.3, shell R=50) given as an object list, 16x9x24 "
                          "clusters, assign_objects_to_clusters only", "baseline_config": "BASELINE.json configs[2], cluster stage alone",
              "lights": args.lights}
    wl = Workload("lights", step, args.lights, 17.0, "k_cluster_walk", config,
                  "lights/sec through assign_objects_to_clusters", "lights/s", kernels=["k_cluster_walk", "k_cluster_fill"])
    wl.keep = (view, keep, lights)
    wl.camera_args = (cam, cfv, fr)
    return wl
