"""Rate of the HOST-pointer entries — the path a Medium binding inside the reference really uses:
gpis_sample_distance_host / gpis_transmittance_host for batch sizes 1 .. 1 Mi, with pageable and with pinned
(gpis_alloc_host) caller memory, next to the device-pointer entry on the same rays.
usage: python tools/host_path_bench.py [--config C1] [--guide 16:32] > gpurun_out/host_path.json"""
import argparse, ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _gpis_pkg
pkg = _gpis_pkg.load_package()


def camera_rays(n, spp=64, width=1920, height=1080, seed=1):
    """scene-S-like primary rays (pinhole at (0,0,4), fov 35, bounding sphere 1.5) in the tile driver's order:
    consecutive records are the spp of one pixel, pixels run along x"""
    rng = np.random.default_rng(seed)
    npix = (n + spp - 1) // spp
    # pixels from the image centre outwards along rows, all inside the sphere's silhouette
    x0, y0 = width // 2 - 200, height // 2 - 100
    px = x0 + (np.arange(npix) % 400)
    py = y0 + (np.arange(npix) // 400) % 200
    px = np.repeat(px, spp)[:n]; py = np.repeat(py, spp)[:n]
    k = np.tile(np.arange(spp), npix)[:n]
    jx, jy = rng.random(n), rng.random(n)
    plane = 1.0 / np.tan(np.radians(35.0) / 2)
    d = np.stack([-1.0 + (px + jx) * 2.0 / width, height / width - (py + jy) * 2.0 / width, -np.full(n, plane)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.array([0.0, 0.0, 4.0])
    b = d @ o
    disc = b * b - (o @ o - 1.5 ** 2)
    sq = np.sqrt(np.maximum(disc, 0))
    r = np.zeros(n, dtype=pkg.RAY_IN)
    r["pos"] = o.astype(np.float32); r["dir"] = d.astype(np.float32)
    r["near_t"] = (-b - sq).astype(np.float32); r["far_t"] = (-b + sq).astype(np.float32)
    r["pixel"][:, 0] = px; r["pixel"][:, 1] = py; r["spp"] = k
    r["scene_seed"] = 0xBA5EBA11
    r["u_jitter"] = rng.random(n).astype(np.float32)
    r["first_scatter"] = 1
    return r


def pinned_array(lib, dtype, n):
    p = lib.gpis_alloc_host(dtype.itemsize * n)
    assert p, "gpis_alloc_host failed"
    buf = (ctypes.c_char * (dtype.itemsize * n)).from_address(p)
    return np.frombuffer(buf, dtype=dtype, count=n), p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1")
    ap.add_argument("--guide", default="16:32")
    ap.add_argument("--max-log2", type=int, default=20)
    args = ap.parse_args()
    import torch
    params = pkg.params_for_config(args.config)
    med = pkg.Medium(params)
    lib = med.L.lib
    if args.guide != "off" and int(med.derived()["fast_path"]):
        h, p = (int(x) for x in args.guide.split(":"))
        med.build_guide(h, p)
    nmax = 1 << args.max_log2
    rays_all = camera_rays(nmax)
    pin_in, p_in = pinned_array(lib, pkg.RAY_IN, nmax)
    pin_out, p_out = pinned_array(lib, pkg.SEG_OUT, nmax)
    pin_in[:] = rays_all
    out_pg = np.zeros(nmax, dtype=pkg.SEG_OUT)
    d_in = torch.from_numpy(rays_all.view(np.uint8).reshape(-1).copy()).cuda()
    d_out = torch.zeros(nmax * pkg.SEG_OUT.itemsize, dtype=torch.uint8, device="cuda")
    res = {"config": args.config, "guide": args.guide, "unit": "segments/s (sampleDistance)", "rows": []}
    sizes = [1, 16, 64, 1024, 16384, 65536, 1 << 18, nmax]
    for n in [s for s in sizes if s <= nmax]:
        reps = max(2, min(200, (1 << 18) // n))
        row = {"n": n}
        for name, src, dst in (("pageable", rays_all, out_pg), ("pinned", pin_in, pin_out)):
            fn = lambda: med.L.check(lib.gpis_sample_distance_host(med.h, ctypes.c_size_t(n), src.ctypes.data_as(ctypes.c_void_p), dst.ctypes.data_as(ctypes.c_void_p), None), "host")
            fn(); fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            dt = (time.perf_counter() - t0) / reps
            row[name + "_seg_per_s"] = n / dt
            row[name + "_us_per_call"] = dt * 1e6
        s = torch.cuda.current_stream().cuda_stream
        fn = lambda: med.call("gpis_sample_distance_batch", ctypes.c_size_t(n), d_in.data_ptr(), d_out.data_ptr(), None, s)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row["device_seg_per_s"] = n / dt
        row["device_us_per_call"] = dt * 1e6
        # the host path returns the bytes of the device path
        got = d_out.cpu().numpy().view(pkg.SEG_OUT)[:n]
        row["same_bytes"] = bool(np.array_equal(got.view(np.uint8), pin_out[:n].view(np.uint8)) and np.array_equal(got.view(np.uint8), out_pg[:n].view(np.uint8)))
        res["rows"].append(row)
        print(json.dumps(row), file=sys.stderr)
    lib.gpis_free_host(p_in); lib.gpis_free_host(p_out)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
