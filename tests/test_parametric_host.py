"""Host side of solveGraphParametric (rome_jl_amd/parametric.py) on the CPU: the fixed sparsity patterns computed once per problem -- the CSR slots
of the Jacobian and the blockwise normal matrix -- against scipy's own conversions, and the BLAS thread limit the solve runs under."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rome_jl_amd as R   # noqa: E402
from rome_jl_amd import api, parametric as PM   # noqa: E402


def _random_blocks(P, rng):
    vals, blocks = [], []
    for k, g in P.groups.items():
        dz, dr, da, db = api._LIN_DIMS[k]
        F = len(g["a"])
        Ja = rng.standard_normal((F, dr, da))
        Jb = rng.standard_normal((F, dr, db)) if db else None
        vals.append(Ja.ravel())
        if db:
            vals.append(Jb.ravel())
        blocks.append((Ja, Jb))
    return np.concatenate(vals), blocks


@pytest.mark.parametrize("graph", ["helix", "hexagon", "manhattan"])
def test_fixed_patterns_equal_scipys_conversions(graph):
    import scipy.sparse as sp
    if graph == "helix":
        fg = R.synth_helix3d(P=200, N=8)                       # Pose3Pose3 + PriorPose3: 6 x 6 blocks
    elif graph == "hexagon":
        fg = R.generateGraph_Hexagonal(N=16)                    # Pose2Pose2 + a landmark with bearing-range sightings + priors
    else:
        fg = R.loadG2o(os.path.join(os.path.dirname(__file__), "golden", "manhattan.g2o"), N=8, max_edges=400)
    P = PM._Problem(fg)
    vals, blocks = _random_blocks(P, np.random.default_rng(3))
    J0 = sp.csr_matrix((vals, (P.rows, P.cols_p)), shape=(P.m, P.n))                                   # scipy's COO -> CSR
    J = sp.csr_matrix((vals[P.csr_perm], P.csr_indices, P.csr_indptr), shape=(P.m, P.n))               # the precomputed slots
    assert abs(J - J0).max() == 0.0
    H0 = (J0.T @ J0).tocsc()
    H = P.normal_matrix(blocks)
    assert H.nnz == H0.nnz and abs(H - H0).max() <= 1e-13 * abs(H0).max()
    lam = 0.37
    Hd0 = H0 + lam * sp.diags(H0.diagonal() + 1e-12)
    assert abs(P.damped(H, lam) - Hd0).max() <= 1e-13 * abs(Hd0).max()
    assert abs(P.damped(H, lam) - H).max() > 0 and abs(P.normal_matrix(blocks) - H).max() == 0.0      # damped() works on a copy


def test_blas_limit_is_a_cached_context_manager():
    import scipy.sparse.linalg   # noqa: F401  (the BLAS libraries are loaded before the controller is built)
    with PM._blas_single_thread():
        pass
    c = PM._BLAS_CONTROLLER[0]
    with PM._blas_single_thread():
        assert PM._BLAS_CONTROLLER[0] is c                      # built once
        if c is not None:
            assert all(lib.num_threads == 1 for lib in c.lib_controllers if lib.user_api == "blas")


def test_gc_paused_restores_the_collector_state():
    import gc
    from rome_jl_amd.graph import gc_paused
    assert gc.isenabled()
    with gc_paused():
        assert not gc.isenabled()
        with gc_paused():                                        # nested: the inner exit must not re-enable inside the outer pause
            assert not gc.isenabled()
        assert not gc.isenabled()
    assert gc.isenabled()
    with pytest.raises(RuntimeError):
        with gc_paused():
            raise RuntimeError("x")
    assert gc.isenabled()
    gc.disable()
    try:
        with gc_paused():
            pass
        assert not gc.isenabled()                                # a caller that runs without the collector keeps running without it
    finally:
        gc.enable()
