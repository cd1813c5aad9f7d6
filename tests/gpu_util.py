"""Helpers for the -m gpu tests: device buffers come from torch (plumbing only); every compute
call goes through the C ABI of libgpis_hip.so."""
import numpy as np


def to_dev(arr):
    import torch
    a = np.ascontiguousarray(arr)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()
    return t


def dev_empty(nbytes):
    import torch
    return torch.zeros(max(int(nbytes), 1), dtype=torch.uint8, device="cuda")


def to_host(t, dtype, shape=None):
    import torch
    torch.cuda.synchronize()
    a = t.cpu().numpy().view(dtype)
    return a.reshape(shape) if shape is not None else a


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def scene_rays(ob, orc, scene, step=1, spp=None):
    """All primary rays of scene S that hit the bounding sphere (oracle ray generator)."""
    import numpy as np
    rays, us = [], []
    h, w = int(scene["height"]), int(scene["width"])
    n_spp = int(scene["spp_count"]) if spp is None else spp
    for y in range(0, h, step):
        for x in range(0, w, step):
            for k in range(n_spp):
                hit, ray, u = orc.scene_s_primary(scene, x, y, int(scene["spp_begin"]) + k)
                if hit:
                    rays.append(ray)
                    us.append(u)
    return np.array(rays, dtype=ob.RAY_IN), np.array(us, dtype=np.float32)


def shadow_rays_from(ob, scene, rays, us, seg):
    """Shadow segments the scene-S estimator would trace for these primary results (numpy
    restatement of the driver's shading step, used to build transmittance inputs)."""
    out = []
    l = np.asarray(scene["light_dir"], dtype=np.float32)
    l = l / np.float32(np.sqrt(np.float32((l * l).sum())))
    R = float(scene["bound_radius"])
    for r, u, o in zip(rays, us, seg):
        if not o["ok"] or o["exited"]:
            continue
        a = o["aniso"]
        n = (a / np.sqrt((a * a).sum())).astype(np.float32)
        if not float(np.dot(n, l)) > 0:
            continue
        p = o["p"].astype(np.float64)
        d = l.astype(np.float64)
        b = float(np.dot(p, d))
        c = float(np.dot(p, p)) - R * R
        disc = b * b - float(np.dot(d, d)) * c
        if disc <= 0:
            continue
        t1 = (-b + np.sqrt(disc)) / float(np.dot(d, d))
        if t1 <= 0:
            continue
        s = np.zeros((), dtype=ob.RAY_IN)
        s["pos"] = o["p"]
        s["dir"] = l
        s["near_t"] = 0
        s["far_t"] = np.float32(t1)
        s["pixel"] = r["pixel"]
        s["spp"] = r["spp"]
        s["segment"] = r["segment"] + 1
        s["scene_seed"] = r["scene_seed"]
        s["info_t"] = r["info_t"] + o["sample_t"]
        s["u_jitter"] = u
        s["first_scatter"] = 0
        s["bounce"] = r["bounce"] + 1
        s["last_val"] = o["last_val"]
        s["last_gp_id"] = o["gp_id"]
        s["last_aniso"] = o["aniso"]
        out.append(s)
    return np.array(out, dtype=ob.RAY_IN)
