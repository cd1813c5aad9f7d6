"""Hit/miss flips between the HIP path and the CPU restatement on the toleranced configurations (double libm on the device is
ocml, on the CPU glibc): scene-S primary segments + the shadow segments they spawn.  One JSON object on stdout."""
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


def main():
    import oracle_bindings as ob
    from gpu_util import scene_rays, shadow_rays_from
    cases = {}

    def case(name, params, w, h, step, spp=1):
        med, orc = pkg.Medium(params), ob.Oracle(params, threads=min(os.cpu_count() or 1, 64))
        scene = ob.default_scene_s(w, h, spp)
        rays, us = scene_rays(ob, orc, scene, step=step)
        want = orc.sample_distance(rays)
        sh = shadow_rays_from(ob, scene, rays, us, want)
        batch = np.concatenate([rays, sh]) if len(sh) else rays
        got, want = med.sample_distance(batch), orc.sample_distance(batch)
        vis_g, vis_o = med.transmittance(batch), orc.transmittance(batch)
        same = got["exited"] == want["exited"]
        dt = np.abs(got["t"][same] - want["t"][same])
        cases[name] = {"segments": int(len(batch)), "sample_distance_flips": int((~same).sum()), "transmittance_flips": int((vis_g != vis_o).sum()),
                       "max_abs_dt": float(dt.max()) if len(dt) else 0.0}
        print(name, cases[name], file=sys.stderr, flush=True)

    if "--mid" in sys.argv:
        # mid-size frames of the two toleranced BASELINE configurations: every sample of a 240x135x4 frame (129 600 primary segments
        # + the shadow segments they spawn), minutes of oracle time on the box's host cores
        case("C2 240x135x4 (1D sampling, MIS, Renewal+, rho 32)", pkg.params_for_config("C2"), 240, 135, 1, 4)
        case("C3 240x135x4 (multi-resolution, rho 64, per-path realizations)", pkg.params_for_config("C3"), 240, 135, 1, 4)
        print(json.dumps({"what": "hit/miss decisions that differ between libgpis_hip.so and the CPU restatement (same rays, same seeds), mid-size frames", "cases": cases}, indent=1))
        return
    p = pkg.params_for_config("C2")
    case("C2 (1D sampling, MIS, Renewal+, rho 32)", p, 256, 144, 2)
    p = pkg.params_for_config("C3")
    case("C3 (multi-resolution, rho 64, per-path realizations)", p, 128, 72, 2)
    p = pkg.params_for_config("C3"); p["impulse_density"] = 16; p["multi_resolution_grid"] = 0; p["isotropic_3d_sampling"] = 0
    case("proc_nonstationary ls ramp, brute force, world space, rho 16", p, 192, 108, 2)
    p = pkg.params_for_config("C3"); p["impulse_density"] = 12
    p["var"]["enabled"], p["var"]["type"] = 1, 0
    p["var"]["min"], p["var"]["max"], p["var"]["start"], p["var"]["end"] = 0.4, 1.8, -1.0, 1.0
    p["aniso_field"]["enabled"], p["aniso_field"]["type"] = 1, 0
    p["aniso_field"]["min"], p["aniso_field"]["max"], p["aniso_field"]["start"], p["aniso_field"]["end"] = 0.1, 0.9, -1.0, 1.0
    case("proc_nonstationary with var and aniso fields, multi-resolution, rho 12", p, 192, 108, 2)
    for typ, nm in ((4, "sandstone"), (5, "rust")):
        p = pkg.params_for_config("C3"); p["impulse_density"] = 12; p["ls_ramp_type"] = typ
        p["var"]["enabled"], p["var"]["type"], p["var"]["min"], p["var"]["max"] = 1, typ, 0.5, 1.6
        case("proc_nonstationary with %s ls and var fields, multi-resolution, rho 12" % nm, p, 128, 72, 2)
    for name, kt, extra in (("Matern v=2.5", 1, {"matern_v": 2.5}), ("Matern v=1.5", 1, {"matern_v": 1.5}), ("Matern v=0.5", 1, {"matern_v": 0.5}),
                            ("Gabor aniso", 2, {"gabor_a_inv": 0.08, "gabor_f_inv": 0.06, "gabor_omega": (0.3, 1.0, -0.2)}),
                            ("Gabor iso", 3, {"gabor_a_inv": 0.08, "gabor_f_inv": 0.06})):
        p = pkg.params_for_config("C0"); p["single_realization"] = 0; p["correlation_context"] = pkg.CTX.RENEWAL; p["impulse_density"] = 12
        p["kernel_type"] = kt
        for k, v in extra.items():
            p[k] = v
        case(name + ", world space, renewal, rho 12", p, 192, 108, 2)
    print(json.dumps({"what": "hit/miss decisions that differ between libgpis_hip.so and the CPU restatement (same rays, same seeds)", "cases": cases}, indent=1))


if __name__ == "__main__":
    main()
