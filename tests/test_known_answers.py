"""Analytic known-answer tests for the arithmetic the reference takes from nalgebra-glm ("parity unpinned"
for a5-a7, a10, a14, a15).  Every expected value here is derived by hand / in float64 from the mathematical definition
(rotation about an axis, T*R*S, plane equations of a perspective frustum, bounding-sphere distance) -- NOT from the oracle,
the device code or the numpy mirror, which were all written from one reading of nalgebra 0.25.  A systematic misreading
(operand order of rotate / translate / scale, the sign of the rotation, row/column of the plane extraction, the near plane
quirk) moves these values by O(1) and fails the tests; rounding differences are far below the tolerance.

The same expectations are checked twice: against the oracle's functions (CPU, `-m "not gpu"`) and against the HIP path
through the C ABI (`-m gpu`).  Tolerance: 1e-5 abs on matrices (north_star), stated at each assert.
"""
import ctypes as C

import numpy as np
import pytest

import oracle as ro
from helpers import to_oracle, oracle_camera

MAT_TOL = 1e-5          # north_star: emitted 4x4 model matrices within 1e-5 abs


# ---------------------------------------------------------------------------------------------------------------
# float64 ground truth from the definitions
# ---------------------------------------------------------------------------------------------------------------
def rotation64(axis, angle):
    """right-handed rotation by `angle` about `axis` (Rodrigues): R = cos I + sin [u]x + (1 - cos) u u^T"""
    u = np.asarray(axis, np.float64); u = u / np.linalg.norm(u)
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.cos(angle) * np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * np.outer(u, u)


def trs64(pos, axis=None, angle=0.0, scale=None):
    """glm: translate(I, p) * rotate(angle, axis) * scale(s), as a column-major 16-vector"""
    M = np.eye(4)
    M[:3, 3] = pos
    if axis is not None:
        R = np.eye(4); R[:3, :3] = rotation64(axis, angle); M = M @ R
    if scale is not None:
        M = M @ np.diag([scale[0], scale[1], scale[2], 1.0])
    return M.T.reshape(16)            # column-major: m[col * 4 + row]


def two_corner_aabb64(orig, m16):
    """StaticAABB::apply_transformation: only the min and the max corner are transformed (aabb.rs:95-114)"""
    M = np.asarray(m16, np.float64).reshape(4, 4).T
    a = M @ np.array([orig[0], orig[2], orig[4], 1.0]); b = M @ np.array([orig[1], orig[3], orig[5], 1.0])
    return np.array([min(a[0], b[0]), max(a[0], b[0]), min(a[1], b[1]), max(a[1], b[1]), min(a[2], b[2]), max(a[2], b[2])])


def frustum_planes64(pos, direction, up, fovy, aspect, far):
    """the six planes (nx, ny, nz, w), unit normals pointing inwards, of a perspective frustum -- in the reference's order
    L, R, B, T, N, F and with its near-plane quirk: N passes through the camera position (render_frustum_culler.rs:76)"""
    d = np.asarray(direction, np.float64); d /= np.linalg.norm(d)
    r = np.cross(d, up); r /= np.linalg.norm(r)
    u = np.cross(r, d)
    ty = np.tan(fovy / 2.0); tx = ty * aspect
    def plane(n):
        n = n / np.linalg.norm(n)
        return np.append(n, -np.dot(n, pos))
    P = [plane(d * tx + r), plane(d * tx - r), plane(d * ty + u), plane(d * ty - u), plane(d)]
    P.append(np.append(-d, np.dot(d, pos) + far))
    return np.array(P)


CASES_TRS = [
    # pos, axis, angle, scale
    ((0, 0, 0), None, 0.0, None),
    ((3, 4, 5), None, 0.0, None),
    ((0, 0, 0), (0, 0, 1), np.pi / 2, None),
    ((0, 0, 0), (1, 0, 0), np.pi / 2, None),
    ((0, 0, 0), (0, 1, 0), np.pi / 2, None),
    ((0, 0, 0), (0, 0, 1), np.pi, None),
    ((0, 0, 0), (0, 1, 0), -np.pi / 2, None),
    ((0, 0, 0), None, 0.0, (2, 3, 4)),
    ((3, 4, 5), (0, 0, 1), np.pi / 2, (2, 3, 4)),              # T*R*S: col0 = (0, 2, 0), col1 = (-3, 0, 0), col3 = (3, 4, 5)
    ((8000.5, 8100.25, 7900.75), (1, 2, 3), 0.7, (0.5, 1.5, 2.5)),
    ((100, 200, 300), (0, 3, 0), 2.5, (2, 2, 2)),               # un-normalised axis
]


def test_hand_written_matrices():
    """a few matrices written out by hand (no helper in between): 90 degrees about z maps x -> y, y -> -x"""
    m = trs64((3, 4, 5), (0, 0, 1), np.pi / 2, (2, 3, 4))
    np.testing.assert_allclose(m, [0, 2, 0, 0, -3, 0, 0, 0, 0, 0, 4, 0, 3, 4, 5, 1], atol=1e-12)
    m = trs64((0, 0, 0), (1, 0, 0), np.pi / 2)                                  # about x: y -> z, z -> -y
    np.testing.assert_allclose(m, [1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1], atol=1e-12)
    m = trs64((0, 0, 0), (0, 1, 0), np.pi / 2)                                  # about y: z -> x, x -> -z
    np.testing.assert_allclose(m, [0, 0, -1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1], atol=1e-12)


@pytest.mark.parametrize("pos,axis,angle,scale", CASES_TRS)
def test_oracle_trs_matrix(pos, axis, angle, scale):
    got = ro.trs_matrix(pos, axis, np.float32(angle), scale)
    want = trs64(pos, axis, float(np.float32(angle)), scale)
    np.testing.assert_allclose(got, want, atol=MAT_TOL, rtol=0)
    if axis is None and scale is None:
        np.testing.assert_array_equal(got, want.astype(np.float32))            # identity / pure translation: exact


def test_oracle_two_corner_aabb():
    L = ro.lib()
    # 90 degrees about z + translation: min corner (-1,-1,-1) -> (1,-1,-1), max corner (1,1,1) -> (-1,1,1)
    m = ro.trs_matrix((10, 20, 30), (0, 0, 1), np.float32(np.pi / 2))
    a = np.array(L.ro_apply_transformation(ro.aabb((-1, 1, -1, 1, -1, 1)), ro._fp(m)).tup())
    np.testing.assert_allclose(a, [9, 11, 19, 21, 29, 31], atol=MAT_TOL)
    # 45 degrees about z: both corners land on x = 0 -- the reference's box degenerates in x (the true box would be +-sqrt 2)
    m = ro.trs_matrix((0, 0, 0), (0, 0, 1), np.float32(np.pi / 4))
    a = np.array(L.ro_apply_transformation(ro.aabb((-1, 1, -1, 1, -1, 1)), ro._fp(m)).tup())
    np.testing.assert_allclose(a, [0, 0, -np.sqrt(2), np.sqrt(2), -1, 1], atol=MAT_TOL)
    # non-uniform scale + rotation, against the float64 two-corner rule
    orig = (-1, 2, -3, 1, 0.5, 4)
    m = ro.trs_matrix((5, 6, 7), (1, 1, 0), np.float32(1.1), (2, 0.5, 3))
    a = np.array(L.ro_apply_transformation(ro.aabb(orig), ro._fp(m)).tup())
    np.testing.assert_allclose(a, two_corner_aabb64(orig, trs64((5, 6, 7), (1, 1, 0), float(np.float32(1.1)), (2, 0.5, 3))), atol=MAT_TOL)


CAMERAS = [
    # pos, dir, fov degrees, aspect (w, h), far
    ((0, 0, 0), (0, 0, -1), 90.0, (720, 720), 100.0),
    ((100, 200, 300), (0, 0, -1), 90.0, (720, 720), 100.0),
    ((100, 200, 300), (1, 0, 0), 45.0, (1280, 720), 1000.0),
    ((8192, 8192, 8192), (0.0, 0.6, -0.8), 60.0, (1280, 720), 2500.0),
]


def camera_pv(pos, d, fov, window, far, near=1.0):
    proj = ro.perspective(np.float32(window[0]) / np.float32(window[1]), np.float32(np.radians(np.float32(fov))), near, far)
    view = ro.look_at(pos, np.asarray(pos, np.float32) + np.asarray(d, np.float32))
    return ro.mat4_mul(proj, view)


def test_perspective_entries():
    """nalgebra Perspective3::new(aspect, fovy, near, far) == the OpenGL projection matrix"""
    p = ro.perspective(np.float32(16 / 9), np.float32(np.radians(45.0)), 0.1, 1000.0).reshape(4, 4).T    # [row][col]
    t = np.tan(np.radians(45.0) / 2)
    want = np.zeros((4, 4)); want[0, 0] = 1 / (t * 16 / 9); want[1, 1] = 1 / t
    want[2, 2] = (1000.0 + 0.1) / (0.1 - 1000.0); want[2, 3] = 2 * 1000.0 * 0.1 / (0.1 - 1000.0); want[3, 2] = -1
    np.testing.assert_allclose(p, want, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("pos,d,fov,window,far", CAMERAS)
def test_oracle_planes(pos, d, fov, window, far):
    got = ro.make_planes(camera_pv(pos, d, fov, window, far))
    want = frustum_planes64(np.asarray(pos, np.float64), d, (0, 1, 0), np.radians(fov), window[0] / window[1], far)
    np.testing.assert_allclose(got[:, :3], want[:, :3], atol=2e-4)       # unit normals (the far plane divides two nearly equal numbers)
    # w by geometry: the camera position lies ON the four side planes and on the near plane (the reference's near plane passes through
    # the camera: Near = row 3 alone), and `far` in front of the far plane.  Tolerances: f32 rounding of P*V at coordinates ~ |pos|
    # (side planes, near), and the cancellation of row3 - row2 whose z coefficient is 2n/(f-n) (far plane: 1 %).
    p64 = np.asarray(pos, np.float64)
    dist = got[:, :3].astype(np.float64) @ p64 + got[:, 3]
    np.testing.assert_allclose(dist[:5], 0.0, atol=1e-5 * (100.0 + np.abs(p64).max()) * 3)
    assert abs(dist[5] - far) < 0.01 * far, (dist[5], far)
    # canonical camera at the origin, 90 degrees, aspect 1: written out
    if pos == (0, 0, 0):
        s = np.sqrt(0.5)
        np.testing.assert_allclose(got, [[s, 0, -s, 0], [-s, 0, -s, 0], [0, s, -s, 0], [0, -s, -s, 0], [0, 0, -1, 0], [0, 0, 1, 100]], atol=2e-3)


def test_oracle_frustum_and_logic_predicates():
    L = ro.lib()
    planes = ro.make_planes(camera_pv((0, 0, 0), (0, 0, -1), 90.0, (720, 720), 100.0)).reshape(24)
    vis = lambda box: bool(L.ro_frustum_aabb_visible(ro._fp(planes), ro.aabb(box)))
    assert vis((-1, 1, -1, 1, -11, -9))                 # straight ahead
    assert not vis((-1, 1, -1, 1, 9, 11))               # behind the camera (near plane through the camera)
    assert vis((-1, 1, -1, 1, -0.5, 0.5))               # straddles the camera plane: some corner in front
    assert not vis((20, 22, -1, 1, -11, -9))            # right of the right plane (x > -z)
    assert vis((9, 11, -1, 1, -11, -9))                 # straddles the right plane
    assert not vis((-1, 1, 20, 22, -11, -9))            # above the top plane
    assert not vis((-1, 1, -1, 1, -120, -110))          # beyond the far plane
    assert vis((-1, 1, -1, 1, -101, -99))               # straddles the far plane
    # all-corners-outside-one-plane is the only rejection: a big box around the frustum apex is visible
    assert vis((-500, 500, -500, 500, -500, 500))
    cam = np.array([10, 20, 30], np.float32)
    inview = lambda box, la: bool(L.ro_logic_aabb_in_view(np.float32(la), ro._fp(cam), ro.aabb(box)))
    assert inview((13, 14, 24, 25, 30, 31), 5.0)        # nearest corner (13, 24, 30): distance exactly 5
    assert not inview((13, 14, 24, 25, 30, 31), 4.999)
    assert not inview((9, 11, 19, 21, 29, 31), 1.0)     # the camera INSIDE the box: nearest CORNER is sqrt(3) away (corner test, not box test)
    assert inview((9, 11, 19, 21, 29, 31), 1.75)


def test_oracle_distance_to_aabb():
    L = ro.lib()
    dist = lambda box, cam: float(L.ro_distance_to_aabb(ro.aabb(box), ro._fp(np.asarray(cam, np.float32))))
    assert abs(dist((0, 2, 0, 2, 0, 2), (11, 1, 1)) - (10 - np.sqrt(3))) < 1e-5             # cube side 2: bounding sphere sqrt(3)
    assert dist((0, 2, 0, 2, 0, 2), (1, 1, 1)) == 0.0                                        # inside: clamped at 0
    assert dist((0, 2, 0, 2, 0, 2), (2.5, 1, 1)) == 0.0                                      # outside the box but inside the sphere
    assert abs(dist((0, 4, 0, 2, 0, 2), (12, 1, 1)) - (10 - np.sqrt(12))) < 1e-5            # the LONGEST side sizes the sphere
    assert abs(dist((0, 2, 0, 2, 0, 8), (1, 1, 104)) - (100 - np.sqrt(48))) < 1e-4


def asteroid_world(rate, angle0=np.radians(0.1), pos=(8224.0, 8224.0, 8224.0)):
    import render_engine_amd as R
    C_ = R._capi
    e = np.zeros(1, R.ENTITY_DT)
    e["id"] = 5; e["model_index"] = 3
    e["flags"] = C_.F_HAS_ROT | C_.F_HAS_ROTVEL | C_.F_HAS_SCALE
    e["original"][0] = (-0.5, 0.5, -0.5, 0.5, -0.5, 0.5)
    e["pos"][0] = pos; e["rot_axis"][0] = (0, 1, 0); e["rot_angle"] = np.float32(angle0); e["scale"][0] = (2, 2, 2)
    e["rotvel_axis"][0] = (0, 1, 0); e["rotvel"] = np.float32(rate); e["rotacc_axis"][0] = (1, 0, 0)
    return e


def test_oracle_one_tick_of_an_asteroid():
    """space_logic asteroid (asteroid.rs:118-123): Rotation((0,1,0), 0.1 deg), VelocityRotation((0,1,0), rate), Scale 2.
    One tick of dt: theta = 0.1 deg + rate * dt about +y (both axis sums re-normalise to +y)."""
    import render_engine_amd as R
    rate, dt = np.radians(17.0), 0.016
    ents = asteroid_world(rate)
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    cam = R.Camera((8224, 8224, 8324), (0, 0, -1), 1000.0)
    oc = oracle_camera(cam)
    w.cull(oc); w.render(oc); w.tick(oc, dt)
    o = w.entity(5)
    theta = float(np.float32(np.radians(0.1))) + float(np.float32(rate)) * float(np.float32(dt))
    np.testing.assert_allclose(o["rot"], [0, 1, 0, theta], atol=1e-6)
    c, s = np.cos(theta), np.sin(theta)
    want = [2 * c, 0, -2 * s, 0, 0, 2, 0, 0, 2 * s, 0, 2 * c, 0, 8224, 8224, 8224, 1]      # T * Ry(theta) * S(2), column-major
    np.testing.assert_allclose(o["mat"], want, atol=MAT_TOL)
    np.testing.assert_allclose(o["aabb"], two_corner_aabb64((-0.5, 0.5, -0.5, 0.5, -0.5, 0.5), want), atol=1e-3)
    w.close()


# ---------------------------------------------------------------------------------------------------------------
# the same expectations through the C ABI on the GPU
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def R():
    import render_engine_amd as R
    return R


def trs_world(R):
    C_ = R._capi
    e = np.zeros(len(CASES_TRS), R.ENTITY_DT)
    e["rot_axis"][:, 0] = 1; e["rotvel_axis"][:, 0] = 1; e["rotacc_axis"][:, 0] = 1; e["scale"][:] = 1
    for i, (pos, axis, angle, scale) in enumerate(CASES_TRS):
        e["id"][i] = i; e["flags"][i] = C_.F_STATIC
        e["original"][i] = (-1, 2, -3, 1, 0.5, 4)
        e["pos"][i] = np.asarray(pos, np.float32) + (0 if max(pos) > 50 else 4096)      # inside the world (the small cases are shifted)
        if axis is not None:
            e["flags"][i] |= C_.F_HAS_ROT; e["rot_axis"][i] = axis; e["rot_angle"][i] = np.float32(angle)
        if scale is not None:
            e["flags"][i] |= C_.F_HAS_SCALE; e["scale"][i] = scale
    return e


@pytest.mark.gpu
def test_gpu_trs_matrices_and_aabbs(R):
    e = trs_world(R)
    p = R.Pipeline(16384, 64)
    assert p.register_model_instances(e) == 0
    for i, (pos, axis, angle, scale) in enumerate(CASES_TRS):
        ppos = e["pos"][i].astype(np.float64)
        want = trs64(ppos, axis, float(np.float32(angle)), scale)
        got = p.read_component(i, R._capi.C_TRANSFORMATION)
        np.testing.assert_allclose(got, want, atol=MAT_TOL, rtol=0, err_msg=f"case {i}")
        box = p.read_component(i, R._capi.C_STATIC_AABB)
        np.testing.assert_allclose(box, two_corner_aabb64((-1, 2, -3, 1, 0.5, 4), want), atol=2e-3, err_msg=f"aabb of case {i}")
    p.close()


@pytest.mark.gpu
def test_gpu_one_tick_of_an_asteroid(R):
    rate, dt = np.radians(17.0), 0.016
    p = R.Pipeline(16384, 64)
    p.register_model_instances(asteroid_world(rate))
    g = p.cull_and_pack(R.Camera((8224, 8224, 8324), (0, 0, -1), 1000.0))
    assert list(g["ids"]) == [5]
    t = p.tick(dt)
    assert t["n_changed"] == 1
    theta = float(np.float32(np.radians(0.1))) + float(np.float32(rate)) * float(np.float32(dt))
    np.testing.assert_allclose(p.read_component(5, R._capi.C_ROTATION), [0, 1, 0, theta], atol=1e-6)
    c, s = np.cos(theta), np.sin(theta)
    want = [2 * c, 0, -2 * s, 0, 0, 2, 0, 0, 2 * s, 0, 2 * c, 0, 8224, 8224, 8224, 1]
    np.testing.assert_allclose(p.read_component(5, R._capi.C_TRANSFORMATION), want, atol=MAT_TOL)
    assert int(p.read_component(5, R._capi.C_FLAGS)[0]) & R._capi.F_HAS_ROTATED
    p.close()


# probes around a camera at a section centre looking down -z, fov 45 degrees, aspect 1, far 1000: offsets in sections (64 units).
# Every probe's section box lies entirely inside or entirely outside the frustum by more than its half diagonal (55.4), so the
# verdict follows from the geometry alone: inside iff |x|, |y| <= tan(22.5 deg) * depth and 0 <= depth <= 1000.
PROBES_IN = [(0, 0, -5), (1, 0, -5), (0, -2, -10), (3, 3, -12), (0, 0, -14), (-2, 1, -9), (0, 3, -13)]
PROBES_OUT = [(4, 0, -5), (0, 0, 5), (0, 0, -17), (0, 6, -10), (-7, 0, -14), (0, -4, -6), (7, 7, -12), (0, 0, 12)]


def probe_world(R):
    probes = PROBES_IN + PROBES_OUT
    e = np.zeros(len(probes), R.ENTITY_DT)
    e["rot_axis"][:, 0] = 1; e["rotvel_axis"][:, 0] = 1; e["rotacc_axis"][:, 0] = 1; e["scale"][:] = 1
    for i, (dx, dy, dz) in enumerate(probes):
        e["id"][i] = i; e["flags"][i] = R._capi.F_STATIC; e["model_index"][i] = 1
        e["original"][i] = (-1, 1, -1, 1, -1, 1)
        e["pos"][i] = (8224 + 64 * dx, 8224 + 64 * dy, 8224 + 64 * dz)
    return e


def check_probe_margins():
    t = np.tan(np.radians(22.5)); cosh = np.cos(np.radians(22.5)); half_diag = 32 * np.sqrt(3)
    for (dx, dy, dz), inside in [(q, True) for q in PROBES_IN] + [(q, False) for q in PROBES_OUT]:
        x, y, depth = 64.0 * dx, 64.0 * dy, -64.0 * dz
        d = [(t * depth - abs(x)) * cosh, (t * depth - abs(y)) * cosh, depth, 1000.0 - depth]      # signed distances to side / near / far planes
        if inside:
            assert min(d) > half_diag, (dx, dy, dz, d)
        else:
            assert min(d) < -half_diag, (dx, dy, dz, d)


def test_probe_margins_are_clear_cut():
    check_probe_margins()


def test_oracle_frustum_probes():
    import render_engine_amd as R
    e = probe_world(R)
    w = ro.World(16384, 64); w.register(to_oracle(e))
    cam = R.Camera((8224, 8224, 8224), (0, 0, -1), 1000.0, fov_degrees=45.0, window_dimensions=(720, 720), near_draw_distance=1.0)
    oc = oracle_camera(cam)
    w.cull(oc)
    o = w.render(oc)
    assert sorted(int(i) for i in o["ids"]) == list(range(len(PROBES_IN)))
    w.close()


@pytest.mark.gpu
def test_gpu_frustum_probes(R):
    e = probe_world(R)
    p = R.Pipeline(16384, 64)
    p.register_model_instances(e)
    cam = R.Camera((8224, 8224, 8224), (0, 0, -1), 1000.0, fov_degrees=45.0, window_dimensions=(720, 720), near_draw_distance=1.0)
    g = p.cull_and_pack(cam)
    assert sorted(int(i) for i in g["ids"]) == list(range(len(PROBES_IN)))
    p.close()


def lod_probe(R, D):
    """one cube of side 2 straight ahead at distance D: distance_to_aabb = D - sqrt(3); LOD bands bracket that value"""
    e = np.zeros(1, R.ENTITY_DT)
    e["rot_axis"][:, 0] = 1; e["rotvel_axis"][:, 0] = 1; e["rotacc_axis"][:, 0] = 1; e["scale"][:] = 1
    e["id"] = 9; e["flags"] = R._capi.F_STATIC; e["model_index"] = 2
    e["original"][0] = (-1, 1, -1, 1, -1, 1); e["pos"][0] = (8224, 8224, 8224 - D)
    d = D - np.sqrt(3.0)
    lod = (np.array([0.0, d - 0.01, d + 0.01], np.float32), np.array([d - 0.01, d + 0.01, 1e9], np.float32))
    cam = R.Camera((8224, 8224, 8224), (0, 0, -1), 1000.0, level_of_views=lod)
    return e, cam


@pytest.mark.parametrize("D", [320.0, 777.0])
def test_oracle_distance_selects_the_bracketing_lod_band(D):
    import render_engine_amd as R
    e, cam = lod_probe(R, D)
    w = ro.World(16384, 64); w.register(to_oracle(e))
    oc = oracle_camera(cam)
    w.cull(oc); o = w.render(oc)
    assert [int(m) for m in o["groups"]["model_index"]] == [2 | (1 << 25)]
    w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("D", [320.0, 777.0])
def test_gpu_distance_selects_the_bracketing_lod_band(R, D):
    e, cam = lod_probe(R, D)
    p = R.Pipeline(16384, 64)
    p.register_model_instances(e)
    g = p.cull_and_pack(cam)
    assert [int(m) for m in g["groups"]["model_index"]] == [2 | (1 << 25)]
    p.close()
