"""rome_jl_amd -- MI355X (gfx950) factor convolutions behind RoME.jl's factor plugin surface.

Scope: ONE hot path of RoME.jl + IncrementalInference.jl -- the per-particle residual + root-find
inside `approxConvBelief` for Pose2Pose2, PriorPose2, Pose2Point2BearingRange and Pose3Pose3 --
as hand-written HIP kernels in librome_mi355.so (C ABI: include/rome_mi355.h).  This package is the
host-side mirror of the reference interface; it contains no compute and no CPU fallback.
"""
from . import _lib
from ._lib import (Context, Opts, RomeError, SOLVER_CLOSED_FORM, SOLVER_NEWTON, SOLVER_NELDER_MEAD, SOLVER_GAUSS_NEWTON,
                   LAYOUT_SOA, LAYOUT_AOS, LAYOUT_AOS_POINTS, MAX_PARTICLES, MAX_PARTICLES_REGISTER)
from .factors import (MvNormal, Normal, Uniform, Pose2, Point2, Pose3, Pose2Pose2, PriorPose2, Pose2Point2BearingRange,
                      Pose3Pose3, PriorPose3, PriorPoint2, Point2Point2, getMeasurementParametric, getPoint, getCoordinates, pack_factor,
                      unpack_factor)
from .api import (linearize, belief_stats, kde_bandwidth, kde_max, manifoldProduct, calcPPE, points_to_coords, coords_to_points, calcFactorResidualTemporary, make_opts, cholesky_lower, default_context,
                  residual_pose2pose2, residual_priorpose2, residual_pose2point2br, residual_pose2point2br_pt,
                  residual_pose3pose3, residual_pose3pose3_pt, residual_priorpose3,
                  conv_pose2pose2, conv_pose2point2br, conv_pose3pose3, sample_priorpose2, sample_priorpose3, sample_priorpoint2)
from .graph import (FactorGraph, initfg, fifoFreeze, isMarginalized, importG2o, parseG2oInstruction, loadG2o, synth_manhattan,
                    synth_manhattan_edges, synth_pose2_tables, synth_helix3d, synth_mit_br, add_synthetic_landmarks, dead_reckon_init_pose3, generateGraph_Circle, generateGraph_Hexagonal,
                    PackedGraph, dead_reckon_init)
from .canonical import (generateGraph_ZeroPose, buildGraphChain, generateGraph_TwoPoseOdo, calcHelix_T, generateGraph_Helix2D,
                        generateGraph_Helix2DSlew, generateGraph_Helix2DSpiral, generateGraph_Boxes2D, generateGraph_Beehive,
                        generateGraph_Honeycomb, synth_beehive_mh, exportG2o, stringG2o, getPPE, setPPE, accumulateFactorMeans)
from .convolution import approxConv, approxConvBelief
from .clique import proposalbeliefs, predictbelief, CliqueBatch, upGibbsCliqueDensity, upGibbsCliqueFrontier
from .serialization import loadDFG, saveDFG, packFactor, unpackFactor, packBelief, unpackBelief
from .parametric import solveGraphParametric, initParametric
from .device import DeviceGraph
from .solve import initAll, initAllOrdered, solveGraph, solveTree
from .tree import BayesTree, TreeSolver
from . import distributed


def build(force=False, verbose=False):
    from . import _build
    return _build.build(force=force, verbose=verbose)
