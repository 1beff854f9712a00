"""Host-side mirror of the RoME.jl factor / variable plugin surface for the hot path.

Same names and argument meaning as the reference (paths relative to the RoME.jl checkout):
  variable types  Pose2, Point2, Pose3                       src/variables/VariableTypes.jl:13,35,47
  Pose2Pose2(Z)                                              src/factors/Pose2D.jl:30-32
  PriorPose2(Z)                                              src/factors/PriorPose2.jl:13-15
  Pose2Point2BearingRange(bearing, range)                    src/factors/BearingRange2D.jl:10-13
  Pose3Pose3(Z), PriorPose3(Z)                               src/factors/Pose3Pose3.jl:9-11, Pose3D.jl:8-10
  getMeasurementParametric(::Pose2Point2BearingRange)        src/factors/BearingRange2D.jl:30-37
  Packed* <-> factor converters                              src/factors/Pose2D.jl:76-84 etc.
These objects only hold the measurement model; all arithmetic is done by the HIP library.
"""
import numpy as np


# ---- minimal stand-ins for Distributions.jl beliefs the hot-path factors are built from ----
class MvNormal:
    """MvNormal(μ, Σ) -- Σ may be a full matrix or a vector of the diagonal (like Diagonal(...))."""

    def __init__(self, mu, cov=None):
        mu = np.asarray(mu, dtype=np.float64)
        if cov is None:  # MvNormal(Σ) zero-mean form (Pose2D.jl:31 default)
            cov = mu
            cov = np.asarray(cov, dtype=np.float64)
            mu = np.zeros(cov.shape[0])
        cov = np.asarray(cov, dtype=np.float64)
        if cov.ndim == 1:
            cov = np.diag(cov)
        if cov.shape != (mu.size, mu.size):
            raise ValueError("MvNormal: covariance shape %s does not match mean of length %d" % (cov.shape, mu.size))
        self.mu = mu
        self.cov = cov

    def __repr__(self):
        return "FullNormal(dim: %d μ: %s Σ: %s)" % (self.mu.size, self.mu.tolist(), self.cov.tolist())


class Normal:
    def __init__(self, mu=0.0, sigma=1.0):
        if not sigma >= 0:
            raise ValueError("Normal: sigma must be non-negative")
        self.mu = float(mu)
        self.sigma = float(sigma)

    def __repr__(self):
        return "Normal(μ=%r, σ=%r)" % (self.mu, self.sigma)


class Uniform:
    """Uniform(a, b) (Distributions.jl); accepted for the bearing / range beliefs of Pose2Point2BearingRange."""

    def __init__(self, a, b):
        if not b > a:
            raise ValueError("Uniform: need a < b")
        self.a, self.b = float(a), float(b)
        self.mu = 0.5 * (self.a + self.b)
        self.sigma = -0.5 * (self.b - self.a)   # library encoding: negative "sigma" = half-width of a uniform

    def __repr__(self):
        return "Uniform(a=%r, b=%r)" % (self.a, self.b)


# ---- variable types ----
class _VarType:
    dim = 0        # manifold dimension (tangent coordinates)
    point_len = 0  # doubles in the reference's native point representation
    name = ""

    def __repr__(self):
        return self.name


class _Pose2(_VarType):
    dim, point_len, name = 3, 6, "Pose2"


class _Point2(_VarType):
    dim, point_len, name = 2, 2, "Point2"


class _Pose3(_VarType):
    dim, point_len, name = 6, 12, "Pose3"


Pose2, Point2, Pose3 = _Pose2(), _Point2(), _Pose3()


# ---- factors ----
class _RelativeFactor:
    """<: IIF.AbstractManifoldMinimize"""
    is_prior = False


class _PriorFactor:
    """<: IIF.AbstractPrior"""
    is_prior = True


class Pose2Pose2(_RelativeFactor):
    variable_types = (Pose2, Pose2)

    def __init__(self, Z=None):
        self.Z = Z if Z is not None else MvNormal(np.zeros(3), np.eye(3))  # Pose2D.jl:31
        if self.Z.mu.size != 3:
            raise ValueError("Pose2Pose2 needs a 3-dimensional belief")


class PriorPose2(_PriorFactor):
    variable_types = (Pose2,)

    def __init__(self, Z=None):
        self.Z = Z if Z is not None else MvNormal(np.zeros(3), np.diag([1, 1, 0.1]))  # PriorPose2.jl:14
        if self.Z.mu.size != 3:
            raise ValueError("PriorPose2 needs a 3-dimensional belief")


class Pose2Point2BearingRange(_RelativeFactor):
    variable_types = (Pose2, Point2)

    def __init__(self, bearing, range):  # noqa: A002  (mirrors the reference field name)
        if not isinstance(bearing, (Normal, Uniform)) or not isinstance(range, (Normal, Uniform)):
            raise TypeError("Pose2Point2BearingRange: this build supports Normal / Uniform bearing and range beliefs")
        self.bearing = bearing
        self.range = range


class Pose3Pose3(_RelativeFactor):
    variable_types = (Pose3, Pose3)

    def __init__(self, Z=None):
        self.Z = Z if Z is not None else MvNormal(np.zeros(6), np.diag([0.01] * 3 + [0.0001] * 3))  # Pose3Pose3.jl:10
        if self.Z.mu.size != 6:
            raise ValueError("Pose3Pose3 needs a 6-dimensional belief")


class PriorPose3(_PriorFactor):
    variable_types = (Pose3,)

    def __init__(self, Z=None):
        self.Z = Z if Z is not None else MvNormal(np.zeros(6), np.diag([0.01] * 3 + [0.0001] * 3))  # Pose3D.jl:9
        if self.Z.mu.size != 6:
            raise ValueError("PriorPose3 needs a 6-dimensional belief")


class PriorPoint2(_PriorFactor):
    """Direction observation information of a `Point2` variable (src/factors/Point2D.jl:9-18).  Only the
    parametric path uses it here (it anchors landmarks in the reference's parametric tests)."""
    variable_types = (Point2,)

    def __init__(self, Z=None):
        self.Z = Z if Z is not None else MvNormal(np.zeros(2), np.diag([0.01, 0.01]))  # Point2D.jl:10
        if self.Z.mu.size != 2:
            raise ValueError("PriorPoint2 needs a 2-dimensional belief")


class Point2Point2(_RelativeFactor):
    """Translation between two `Point2` variables (src/factors/Point2D.jl:23-43).  Host-side type only: the box
    generator builds graphs with it; it is not one of the hot-path factor kinds, so packing a graph that holds one for
    the device raises."""
    variable_types = (Point2, Point2)

    def __init__(self, Z=None):
        self.Z = Z if Z is not None else MvNormal(np.zeros(2), np.eye(2))
        if self.Z.mu.size != 2:
            raise ValueError("Point2Point2 needs a 2-dimensional belief")


def getMeasurementParametric(f):
    """(μ, iΣ) as IIF.getMeasurementParametric; BearingRange override at BearingRange2D.jl:30-37."""
    if isinstance(f, Pose2Point2BearingRange):
        if not isinstance(f.bearing, Normal) or not isinstance(f.range, Normal):
            raise TypeError("getMeasurementParametric(::Pose2Point2BearingRange{<:Normal,<:Normal}) only (BearingRange2D.jl:30)")
        return (np.array([f.bearing.mu, f.range.mu]),
                np.diag([1.0 / f.bearing.sigma ** 2, 1.0 / f.range.sigma ** 2]))
    return f.Z.mu.copy(), np.linalg.inv(f.Z.cov)


# ---- Packed* serialisation (dict form of the reference's Packed structs) ----
def _pack_belief(b):
    if isinstance(b, Normal):
        return {"_type": "Normal", "mu": b.mu, "sigma": b.sigma}
    if isinstance(b, Uniform):
        return {"_type": "Uniform", "a": b.a, "b": b.b}
    return {"_type": "FullNormal", "mu": b.mu.tolist(), "cov": b.cov.tolist()}


def _unpack_belief(d):
    if d["_type"] == "Normal":
        return Normal(d["mu"], d["sigma"])
    if d["_type"] == "Uniform":
        return Uniform(d["a"], d["b"])
    return MvNormal(d["mu"], np.asarray(d["cov"]))


def pack_factor(f):
    name = type(f).__name__
    if isinstance(f, Pose2Point2BearingRange):  # PackedPose2Point2BearingRange: bearstr, rangstr (BearingRange2D.jl:76-79)
        return {"fnctype": name, "bearstr": _pack_belief(f.bearing), "rangstr": _pack_belief(f.range)}
    return {"fnctype": name, "Z": _pack_belief(f.Z)}


def unpack_factor(d):
    t = d["fnctype"]
    if t == "Pose2Point2BearingRange":
        return Pose2Point2BearingRange(_unpack_belief(d["bearstr"]), _unpack_belief(d["rangstr"]))
    cls = {"Pose2Pose2": Pose2Pose2, "PriorPose2": PriorPose2, "Pose3Pose3": Pose3Pose3, "PriorPose3": PriorPose3,
           "PriorPoint2": PriorPoint2}[t]
    return cls(_unpack_belief(d["Z"]))


# ---- point <-> coordinate layout helpers (host-side format conversion only) ----
def getPoint(vartype, coords):
    """exp_ϵ(hat(c)) in the reference's native point layout (column-major R)."""
    c = np.asarray(coords, dtype=np.float64)
    if vartype is Pose2:
        s, co = np.sin(c[..., 2]), np.cos(c[..., 2])
        return np.stack([c[..., 0], c[..., 1], co, s, -s, co], axis=-1)
    if vartype is Point2:
        return c.copy()
    if vartype is Pose3:
        from scipy.spatial.transform import Rotation as Rot
        R = Rot.from_rotvec(c[..., 3:].reshape(-1, 3)).as_matrix()  # (n,3,3)
        Rcm = np.transpose(R, (0, 2, 1)).reshape(c.shape[:-1] + (9,))
        return np.concatenate([c[..., :3], Rcm], axis=-1)
    raise TypeError(vartype)


def getCoordinates(vartype, pt):
    """vee(log(ϵ, p))"""
    p = np.asarray(pt, dtype=np.float64)
    if vartype is Pose2:
        return np.stack([p[..., 0], p[..., 1], np.arctan2(p[..., 3], p[..., 2])], axis=-1)
    if vartype is Point2:
        return p.copy()
    if vartype is Pose3:
        from scipy.spatial.transform import Rotation as Rot
        R = np.transpose(p[..., 3:].reshape(-1, 3, 3), (0, 2, 1))
        w = Rot.from_matrix(R).as_rotvec().reshape(p.shape[:-1] + (3,))
        return np.concatenate([p[..., :3], w], axis=-1)
    raise TypeError(vartype)
