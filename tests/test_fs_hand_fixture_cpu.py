"""A hand-checkable anchor for the function-space path: three sample points, no conditioning, a mean far from the
threshold.  Everything the segment does is then a 3x3 problem that fits in a dozen lines of numpy written from the REFERENCE's
description, not from the restatement: PCG32 draws, Box-Muller, one truncated-normal draw that is thrown away
(FunctionSpaceGaussianProcessMedium.cpp:139-148), K = sigma^2 exp(-d^2 / (2 l^2)) on the three points, values = mean + chol(K) z
(Gaussian.cpp:121-167, 179-232).  The CPU restatement must give those values; tests/test_gpu_fs.py holds the device to the
restatement bit for bit, and the GPU variant below repeats this check through the library."""
import numpy as np
import pytest

M64 = (1 << 64) - 1


class Pcg32:
    """UniformSampler (UniformSampler.hpp:41-75): 64-bit LCG, xorshift-rotate output, next1D() from the top 23 bits"""
    def __init__(self, state):
        self.s = int(state)

    def next_u32(self):
        old = self.s
        self.s = (old * 6364136223846793005 + 1) & M64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF

    def next1d(self):
        bits = np.uint32((self.next_u32() >> 9) | 0x3F800000)
        return float(bits.view(np.float32) - np.float32(1.0))


def rand_normal_2(g):
    """Gaussian.cpp:21-34 (PI is the float constant of Angle.hpp:8)"""
    u1, u2 = g.next1d(), g.next1d()
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    ang = float(np.float32(2) * np.float32(3.1415926536)) * u2
    return r * np.cos(ang), r * np.sin(ang)


def expected_segment(params, ray, state):
    n = int(params["fs_sample_points"])
    g = Pcg32(state)
    t_offset = g.next1d()
    ro = ray["pos"].astype(np.float64)
    rd = ray["dir"].astype(np.float64)
    rd = rd / np.sqrt((rd * rd).sum())
    near, far = float(ray["near_t"]), float(ray["far_t"])
    step = (far - near) / n
    max_dist = step * n
    ts = []
    for i in range(n):
        r = min(max((i - t_offset) / (n - 1), 0.0), 1.0)
        a, b = near + step * 0.1, near + max_dist
        rt = a * (1.0 - r) + b * r
        if i == 0:
            rt = near + step * 0.1
        elif i == n - 1:
            rt = near + max_dist
        ts.append(rt)
    pts = np.array([ro + rd * t for t in ts])
    mean = np.full(n, float(params["mean"]["offset"]))              # homogeneous mean
    # sample_start_value: one truncated normal at the segment start, drawn and not used; with a = 0 far below the mean the
    # first Box-Muller pair is accepted (Gaussian.cpp:57-85)
    z1, _ = rand_normal_2(g)
    assert mean[0] + float(params["sigma"]) * z1 >= 0
    s2 = float(np.float32(params["sigma"]) * np.float32(params["sigma"]))
    l2 = float(np.float32(params["length_scale"]) * np.float32(params["length_scale"]))
    d = pts[:, None, :] - pts[None, :, :]
    K = s2 * np.exp(-(d * d).sum(axis=2) / (2 * l2))
    L = np.linalg.cholesky(K)
    z = np.zeros(n)
    for i in range(n // 2):
        z[2 * i], z[2 * i + 1] = rand_normal_2(g)
    if n % 2:
        z[n - 1], _ = rand_normal_2(g)
    return pts, mean + L @ z, near + max_dist


def _case(pkg):
    p = pkg.params_for_config("C0")
    p["single_realization"] = 0
    p["correlation_context"] = pkg.CTX.NONE
    p["mean"]["type"], p["mean"]["offset"] = pkg.MEAN_TYPE.HOMOGENEOUS, 1.0
    p["sigma"], p["length_scale"] = 0.1, 0.05
    p["fs_sample_points"], p["fs_step_size"] = 3, 0.0
    rng = np.random.default_rng(17)
    rays = np.zeros(16, dtype=pkg.RAY_IN)
    rays["pos"] = rng.uniform(-0.4, 0.4, (16, 3))
    d = rng.standard_normal((16, 3))
    rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["near_t"], rays["far_t"] = 0.0, rng.uniform(0.06, 0.12, 16)
    rays["first_scatter"] = 1
    st = np.zeros(16, dtype=pkg.FS_STATE)
    st["sampler_state"] = rng.integers(1, 2**63, size=16, dtype=np.uint64)
    return p, rays, st


def _check(pkg, p, rays, st, out, st_out):
    for i in range(len(rays)):
        pts, vals, max_t = expected_segment(p, rays[i], st["sampler_state"][i])
        assert out["exited"][i] == 1 and out["ok"][i] == 1 and st_out["n_points"][i] == 3 and st_out["is_intersect"][i] == 0
        assert np.allclose(st_out["points"][i][:3], pts, rtol=0, atol=1e-15)
        # numpy's log / cos / sin / exp / cholesky against glibc and Eigen's LLT: a few ulps of a value near 1
        assert np.allclose(st_out["values"][i][:3], vals, rtol=0, atol=5e-15), (i, st_out["values"][i][:3], vals)
        assert abs(out["t"][i] - max_t) < 1e-15


def test_three_point_segment_equals_the_hand_computation(pkg, ob):
    p, rays, st = _case(pkg)
    out, st_out = ob.Oracle(p).fs_sample_distance(rays, st)
    _check(pkg, p, rays, st, out, st_out)


@pytest.mark.gpu
def test_three_point_segment_on_the_device(pkg):
    p, rays, st = _case(pkg)
    out, st_out = pkg.Medium(p).fs_sample_distance(rays, st)
    _check(pkg, p, rays, st, out, st_out)
