"""`approxConvBelief` / `approxConv` mirror (IIF API as used by the reference:
test/testBasicPose2Conv.jl:25, test/TestPoseAndPoint2Constraints.jl:36,97)."""
import numpy as np

from . import _lib, api
from .factors import Pose2Pose2, PriorPose2, Pose2Point2BearingRange, Pose3Pose3, PriorPose3, PriorPoint2


def approxConv(fg, flabel, target, solver=_lib.SOLVER_NEWTON, seed=None, ctx=None, **optkw):
    """N proposal points (dim, N) for variable `target` through factor `flabel`, convolving the
    factor's measurement model with the current belief of its other variable."""
    _, labels, f = fg.getFactor(flabel)
    if target not in labels:
        raise KeyError("%s is not connected to factor %s" % (target, flabel))
    opts = api.make_opts(N=fg.N, solver=solver, seed=seed, **optkw)
    if isinstance(f, PriorPose2):
        return api.sample_priorpose2(opts, [f.Z.mu], [f.Z.cov], ctx=ctx)[0]
    if isinstance(f, PriorPose3):
        return api.sample_priorpose3(opts, [f.Z.mu], [f.Z.cov], ctx=ctx)[0]
    if isinstance(f, PriorPoint2):
        return api.sample_priorpoint2(opts, [f.Z.mu], [f.Z.cov], ctx=ctx)[0]
    mh = fg.multihypo.get(flabel)
    if mh is not None and isinstance(f, Pose2Pose2):   # Pose2Pose2 over [a, b1, b2]
        a, b1, b2 = labels
        opts.layout = _lib.LAYOUT_SOA
        z3 = lambda l: fg.getVal(l)[None] if fg.isInitialized(l) else np.zeros((1, 3, fg.N))
        if target == a:
            return api.conv_pose2pose2(opts, [f.Z.mu], [f.Z.cov], z3(b1), z3(a), dirs=[1], alt=z3(b2), hypo_w=[mh[0]], ctx=ctx)[0]
        prim, alt, w = (b1, b2, mh[0]) if target == b1 else (b2, b1, mh[1])
        return api.conv_pose2pose2(opts, [f.Z.mu], [f.Z.cov], fg.getVal(a)[None], z3(prim), dirs=[0], alt=z3(alt), hypo_w=[w], ctx=ctx)[0]
    if mh is not None:   # Pose2Point2BearingRange over [pose, l1, l2]
        pose, l1, l2 = labels
        opts.layout = _lib.LAYOUT_SOA
        z2 = lambda l: fg.getVal(l)[None] if fg.isInitialized(l) else np.zeros((1, 2, fg.N))
        args = ([[f.bearing.mu, f.range.mu]], [[f.bearing.sigma, f.range.sigma]])
        if target == pose:
            u0 = fg.getVal(pose)[None] if fg.isInitialized(pose) else np.zeros((1, 3, fg.N))
            return api.conv_pose2point2br(opts, 1, *args, z2(l1), u0, alt=z2(l2), hypo_w=[mh[0]], ctx=ctx)[0]
        prim, alt, w = (l1, l2, mh[0]) if target == l1 else (l2, l1, mh[1])
        return api.conv_pose2point2br(opts, 0, *args, fg.getVal(pose)[None], z2(prim), alt=z2(alt), hypo_w=[w], ctx=ctx)[0]
    direction = 0 if labels[1] == target else 1
    other = labels[0] if direction == 0 else labels[1]
    if not fg.isInitialized(other):
        raise ValueError("approxConv: variable %s has no belief yet" % other)
    fixed = fg.getVal(other)[None]
    tt = fg.variables[target]
    u0 = fg.getVal(target)[None] if fg.isInitialized(target) else np.zeros((1, tt.dim, fg.N))
    if isinstance(f, Pose2Pose2):
        return api.conv_pose2pose2(opts, [f.Z.mu], [f.Z.cov], fixed, u0, dirs=[direction], ctx=ctx)[0]
    if isinstance(f, Pose3Pose3):
        return api.conv_pose3pose3(opts, [f.Z.mu], [f.Z.cov], fixed, u0, dirs=[direction], ctx=ctx)[0]
    if isinstance(f, Pose2Point2BearingRange):
        return api.conv_pose2point2br(opts, direction, [[f.bearing.mu, f.range.mu]], [[f.bearing.sigma, f.range.sigma]],
                                      fixed, u0, ctx=ctx)[0]
    raise TypeError("unsupported factor type %s" % type(f).__name__)


approxConvBelief = approxConv
