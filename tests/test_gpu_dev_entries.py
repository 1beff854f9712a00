"""The entry points of include/rome_mi355.h that no other test calls by name: the device-pointer twins of the parametric linearisation and of
getKDEMax, the thin device-memory helpers a caller without a HIP binding uses (the Julia shim), the device count and the HIP error
accessors.  A `_dev` entry must return what its host-pointer twin returns, bit for bit -- the twin IS the same kernel behind a
staging copy."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import rome_jl_amd as R
    from rome_jl_amd import _lib, api
    return torch, R, _lib, api, _lib.load(), R.Context(0)


def test_device_count_and_error_accessors(env):
    torch, R, _lib, api, lib, ctx = env
    assert lib.rome_device_count() == torch.cuda.device_count() >= 1
    assert lib.rome_last_hip_error(ctx.handle) == 0                       # nothing failed on this context
    assert isinstance(lib.rome_last_hip_error_string(ctx.handle).decode(), str)
    assert lib.rome_last_hip_error(None) == 0                             # a NULL context: no error to report, no crash
    h = C.c_void_p()
    assert lib.rome_ctx_create(C.byref(h), 10 ** 6) == _lib.ERR_INVALID_ARG  # no such device: refused, no context made
    assert not h.value
    assert lib.rome_ctx_create(C.byref(h), -1) == _lib.ERR_INVALID_ARG      # the survey's "device = -1: CPU oracle" is deliberately refused
    assert lib.rome_ctx_create(None, 0) == _lib.ERR_INVALID_ARG


def test_dev_alloc_upload_download_roundtrip(env):
    torch, R, _lib, api, lib, ctx = env
    rng = np.random.default_rng(3)
    src = rng.normal(size=4097)
    back = np.zeros_like(src)
    p = C.c_void_p()
    _lib.check(lib.rome_dev_alloc(ctx.handle, src.nbytes, C.byref(p)), ctx.handle)
    assert p.value
    _lib.check(lib.rome_dev_upload(ctx.handle, p, src.ctypes.data, src.nbytes), ctx.handle)
    _lib.check(lib.rome_dev_download(ctx.handle, back.ctypes.data, p, src.nbytes), ctx.handle)
    assert np.array_equal(src, back)
    # a partial, offset copy: the helpers take plain byte counts and pointers
    back[:] = 0
    _lib.check(lib.rome_dev_download(ctx.handle, back.ctypes.data, C.c_void_p(p.value + 8 * 100), 8 * 50), ctx.handle)
    assert np.array_equal(back[:50], src[100:150]) and not back[50:].any()
    # zero bytes is a no-op, NULL pointers with bytes > 0 are refused
    assert lib.rome_dev_upload(ctx.handle, p, src.ctypes.data, 0) == _lib.OK
    assert lib.rome_dev_upload(ctx.handle, None, src.ctypes.data, 8) == _lib.ERR_INVALID_ARG
    assert lib.rome_dev_download(ctx.handle, back.ctypes.data, None, 8) == _lib.ERR_INVALID_ARG
    assert lib.rome_dev_alloc(ctx.handle, 8, None) == _lib.ERR_INVALID_ARG
    _lib.check(lib.rome_dev_free(ctx.handle, p), ctx.handle)
    assert lib.rome_dev_free(ctx.handle, None) == _lib.OK                 # like free(NULL)


def test_kde_max_dev_equals_the_host_entry(env):
    torch, R, _lib, api, lib, ctx = env
    rng = np.random.default_rng(5)
    V, N = 37, 100
    bel = rng.normal(size=(V, 3, N)) * np.array([2.0, 0.5, 0.3])[None, :, None] + rng.normal(size=(V, 3, 1)) * 5
    bel[:, 2] = (bel[:, 2] + np.pi) % (2 * np.pi) - np.pi
    bw = api.kde_bandwidth(bel, ctx=ctx)
    want = api.kde_max(bel, bw, ctx=ctx)
    d_bel = torch.from_numpy(bel).cuda(); d_bw = torch.from_numpy(bw).cuda(); d_out = torch.zeros((V, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for grid in (0, 200, 64):
        _lib.check(lib.rome_kde_max_dev(ctx.handle, 3, V, N, d_bel.data_ptr(), d_bw.data_ptr(), grid, d_out.data_ptr()), ctx.handle)
        ctx.synchronize()
        ref = want if grid in (0, 200) else api.kde_max(bel, bw, grid_points=grid, ctx=ctx)
        assert np.array_equal(d_out.cpu().numpy(), ref), grid              # 0 = the reference's 200 grid points
    assert lib.rome_kde_max_dev(ctx.handle, 3, V, N, d_bel.data_ptr(), d_bw.data_ptr(), 257, d_out.data_ptr()) == _lib.ERR_INVALID_ARG
    assert lib.rome_kde_max_dev(ctx.handle, 3, V, N, None, d_bw.data_ptr(), 0, d_out.data_ptr()) == _lib.ERR_INVALID_ARG


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4, 5])
def test_linearize_dev_equals_the_host_entry(env, kind):
    torch, R, _lib, api, lib, ctx = env
    dz, dr, da, db = api._LIN_DIMS[kind]
    rng = np.random.default_rng(100 + kind)
    F = 513
    mu = rng.normal(size=(F, dz)); xa = rng.normal(size=(F, da)) * 2; xb = rng.normal(size=(F, db)) * 2 if db else None
    if kind == _lib.FACTOR_POSE2POINT2BR:
        mu[:, 1] = np.abs(mu[:, 1]) + 1.0
    A = rng.normal(size=(F, dr, dr)); W = np.triu(A) + 2 * np.eye(dr)
    r, Ja, Jb = api.linearize(kind, mu, W, xa, xb, ctx=ctx)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    d_mu, d_W, d_xa = t(mu), t(W), t(xa)
    d_xb = t(xb) if db else None
    d_r = torch.zeros((F, dr), dtype=torch.float64, device="cuda"); d_Ja = torch.zeros((F, dr, da), dtype=torch.float64, device="cuda")
    d_Jb = torch.zeros((F, dr, db), dtype=torch.float64, device="cuda") if db else None
    torch.cuda.synchronize()
    ptr = lambda x: x.data_ptr() if x is not None else None          # noqa: E731
    _lib.check(lib.rome_linearize_dev(ctx.handle, kind, F, ptr(d_mu), ptr(d_W), ptr(d_xa), ptr(d_xb), ptr(d_r), ptr(d_Ja), ptr(d_Jb)), ctx.handle)
    ctx.synchronize()
    assert np.array_equal(d_r.cpu().numpy(), r) and np.array_equal(d_Ja.cpu().numpy(), Ja)
    if db:
        assert np.array_equal(d_Jb.cpu().numpy(), Jb)
        assert lib.rome_linearize_dev(ctx.handle, kind, F, ptr(d_mu), ptr(d_W), ptr(d_xa), None, ptr(d_r), ptr(d_Ja), ptr(d_Jb)) == _lib.ERR_INVALID_ARG
    assert lib.rome_linearize_dev(ctx.handle, 6, F, ptr(d_mu), ptr(d_W), ptr(d_xa), ptr(d_xb), ptr(d_r), ptr(d_Ja), ptr(d_Jb)) == _lib.ERR_INVALID_ARG
    assert lib.rome_linearize_dev(ctx.handle, kind, 0, ptr(d_mu), ptr(d_W), ptr(d_xa), ptr(d_xb), ptr(d_r), ptr(d_Ja), ptr(d_Jb)) == _lib.OK   # nothing to do
