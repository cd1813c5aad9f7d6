"""Function-space path (C4), device against the CPU restatement, field by field and bit by bit: how many segments differ in any
output or state word, for the configurations of tests/test_gpu_fs.py (first segment, second segment, shadow segment).  With the
device's exp / log / sincos equal to the host's (csrc/gpis_libm.hpp) the expected count is 0.  One JSON object on stdout."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gpis_pkg  # noqa: E402

pkg = _gpis_pkg.load_package()


def bits_differ(a, b):
    """per record: does any byte differ"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return (a.view(np.uint8).reshape(len(a), -1) != b.view(np.uint8).reshape(len(b), -1)).any(axis=1)


def state_differs(sa, sb):
    """state records compared over their live prefix (n_points / n_values entries), the rest is scratch"""
    out = np.zeros(len(sa), dtype=bool)
    for f in sa.dtype.names:
        x, y = sa[f], sb[f]
        if x.ndim == 1:
            out |= x != y if x.dtype.kind != "f" else x.view("u%d" % x.dtype.itemsize) != y.view("u%d" % y.dtype.itemsize)
            continue
        live = sa["n_values"] if f in ("values",) else sa["n_points"]
        width = x.shape[1]
        per = x.reshape(len(x), width, -1)
        pery = y.reshape(len(y), width, -1)
        k = np.arange(width)[None, :] < np.minimum(live, width)[:, None]
        d = (per.view(np.uint8).reshape(len(x), width, -1) != pery.view(np.uint8).reshape(len(y), width, -1)).any(axis=2)
        out |= (d & k).any(axis=1)
    return out


def main():
    import oracle_bindings as ob
    import test_gpu_fs as T
    cases = [("NONE", 12, 0.0, 0.0), ("RENEWAL", 16, 0.04, 0.05), ("RENEWAL_PLUS", 14, 0.0, 0.0), ("GLOBAL", 14, 0.05, 0.1), ("GLOBAL", 17, 0.0, 1.0),
             ("NONE", 32, 0.0, 0.0), ("RENEWAL_PLUS", 64, 0.01, 0.05), ("GLOBAL", 64, 0.01, 0.1)]
    report = {}
    for ctx, n, step, offset in cases:
        params = T._params(pkg, ctx, n, step, offset, aniso=(1.0, 0.7, 1.4))
        med, orc = pkg.Medium(params), ob.Oracle(params, threads=16)
        rays, st = T._rays(pkg, 384, seed=11 + n)
        got, st_g = med.fs_sample_distance(rays, st)
        want, st_o = orc.fs_sample_distance(rays, st)
        first = bits_differ(got, want) | state_differs(st_g, st_o)
        ok = want["ok"] == 1
        r2 = rays[ok].copy()
        r2["pos"] = rays["pos"][ok] + rays["dir"][ok] * want["sample_t"][ok][:, None]
        rng = np.random.default_rng(5)
        d = rng.standard_normal((len(r2), 3))
        r2["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
        r2["near_t"], r2["far_t"] = 0.0, 0.4
        r2["first_scatter"], r2["bounce"] = 0, 1
        r2["last_aniso"] = want["aniso"][ok]
        got2, st_g2 = med.fs_sample_distance(r2, st_o[ok])
        want2, st_o2 = orc.fs_sample_distance(r2, st_o[ok])
        second = bits_differ(got2, want2) | state_differs(st_g2, st_o2)
        vis_g, sv_g = med.fs_transmittance(r2, st_o[ok])
        vis_o, sv_o = orc.fs_transmittance(r2, st_o[ok])
        shadow = (vis_g != vis_o) | state_differs(sv_g, sv_o)
        key = "%s n=%d step=%g offset=%g" % (ctx, n, step, offset)
        report[key] = {"segments": int(len(rays)), "first_differ": int(first.sum()), "second_segments": int(len(r2)), "second_differ": int(second.sum()),
                       "shadow_differ": int(shadow.sum()), "hits_first": int((want["exited"] == 0).sum())}
        print(key, report[key], file=sys.stderr, flush=True)
    print(json.dumps({"what": "function-space path: segments whose outputs or state differ in any bit between libgpis_hip.so and the CPU restatement", "cases": report}, indent=1))


if __name__ == "__main__":
    main()
