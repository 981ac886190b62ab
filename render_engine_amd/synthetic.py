"""Deterministic synthetic worlds for the configs of BASELINE.json.

Counter-based RNG: splitmix64(seed ^ (index * 0x9E3779B97F4A7C15)), float = top 24 bits / 2^24,
so every entity is reproducible from its index alone (any rank can generate its own shard).
"""
import numpy as np

from . import _capi
from .pipeline import ENTITY_DT

SEED_LAYOUT, SEED_SPIN, SEED_MIX = 0x5EED0001, 0x5EED0002, 0x5EED0003
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        z = x + _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(seed, index, stream):
    """float32 in [0,1) for (seed, entity index, stream number)"""
    with np.errstate(over="ignore"):
        k = np.uint64(seed) ^ (np.asarray(index, np.uint64) * _GOLD) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
    return ((splitmix64(k) >> np.uint64(40)).astype(np.float32) / np.float32(16777216.0)).astype(np.float32)


def lattice_world(cells_per_axis=216, first_cell=20, atomic=64, index_range=None, spinner_every=0, n_models=8,
                  straddler_fraction=0.0, mover_every=0):
    """Cubic lattice: see box_world."""
    return box_world((cells_per_axis,) * 3, first_cell, atomic, index_range, spinner_every, n_models, straddler_fraction, mover_every)


def sub_box_indices(dims, sub_first, sub_dims):
    """entity indices (== ids) of the sub-box [sub_first, sub_first + sub_dims) of a box_world(dims), in ascending order; sub_first
    is relative to the box's first section (x, z, y order like dims)"""
    nx, nz, ny = dims
    ax = np.arange(sub_first[0], sub_first[0] + sub_dims[0], dtype=np.uint64)
    az = np.arange(sub_first[1], sub_first[1] + sub_dims[1], dtype=np.uint64)
    ay = np.arange(sub_first[2], sub_first[2] + sub_dims[2], dtype=np.uint64)
    return ((ax[:, None, None] * np.uint64(nz) + az[None, :, None]) * np.uint64(ny) + ay[None, None, :]).reshape(-1)


def box_world(dims, first_cell=20, atomic=64, index_range=None, spinner_every=0, n_models=8,
              straddler_fraction=0.0, mover_every=0, indices=None):
    """Config 2/3: one entity per level-0 section of a cubic lattice (uniform spatial-hash fill).

    dims = (nx, nz, ny) sections per axis.  Entity i sits in section (cx,cz,cy) = first_cell +
    unravel(i) (x major, then z, then y: the key order, so a contiguous index range is a
    contiguous section-key range -- the multi-GPU shard), half extent h = 0.5 + 1.5*u0, centre jittered inside the section.  All static,
    model ids round-robin.  spinner_every=k: ids == 0 mod k are non-static asteroids
    (space_logic/solar_system/asteroid.rs:118-123): VelocityRotation about +y, rate in
    [-20,20] deg/s, Rotation((0,1,0), 0.1 deg), Scale 2, h <= 1 so the rotated box stays inside.
    straddler_fraction: that share of entities gets h in [20,40] (shared sections, higher levels).
    mover_every=k: ids == 1 mod k additionally carry a Velocity of up to 30 units/s.
    """
    nx, nz, ny = dims
    n_total = nx * nz * ny
    lo, hi = (0, n_total) if index_range is None else index_range
    idx = np.arange(lo, hi, dtype=np.uint64) if indices is None else np.asarray(indices, np.uint64)   # `indices`: any subset of the box (ids stay those of the full world)
    n = len(idx)
    cx = (idx // np.uint64(nz * ny)).astype(np.int64) + first_cell
    cz = ((idx // np.uint64(ny)) % np.uint64(nz)).astype(np.int64) + first_cell
    cy = (idx % np.uint64(ny)).astype(np.int64) + first_cell
    e = np.zeros(n, ENTITY_DT)
    e["id"] = idx.astype(np.uint32)
    e["model_index"] = (idx % np.uint64(n_models)).astype(np.uint32)
    u0, u1, u2, u3 = (uniform(SEED_LAYOUT, idx, s) for s in range(4))
    h = (np.float32(0.5) + np.float32(1.5) * u0).astype(np.float32)
    flags = np.full(n, _capi.F_STATIC, np.uint32)
    scale = np.ones((n, 3), np.float32)
    if spinner_every:
        spin = (idx % np.uint64(spinner_every)) == 0
        h = np.where(spin, np.float32(0.25) + np.float32(0.25) * u0, h).astype(np.float32)      # scaled by 2 -> <= 1
        flags = np.where(spin, np.uint32(_capi.F_HAS_ROT | _capi.F_HAS_ROTVEL | _capi.F_HAS_SCALE), flags).astype(np.uint32)
        scale[spin] = 2.0
        rate = (np.radians(np.float32(40.0)) * (uniform(SEED_SPIN, idx, 0) - np.float32(0.5))).astype(np.float32)
        rate = np.where(rate == 0, np.float32(0.01), rate)
        e["rotvel"] = np.where(spin, rate, 0).astype(np.float32)
        e["rotvel_axis"][:, 1] = 1.0
        e["rot_axis"][:, 1] = 1.0
        e["rot_angle"] = np.where(spin, np.radians(np.float32(0.1)), 0).astype(np.float32)
    else:
        e["rot_axis"][:, 0] = 1.0; e["rotvel_axis"][:, 0] = 1.0
    e["rotacc_axis"][:, 0] = 1.0
    if straddler_fraction > 0:
        big = uniform(SEED_MIX, idx, 0) < np.float32(straddler_fraction)
        if spinner_every:
            big &= ~spin
        h = np.where(big, np.float32(20.0) + np.float32(20.0) * u0, h).astype(np.float32)
    if mover_every:
        mv = (idx % np.uint64(mover_every)) == 1
        if spinner_every:
            mv &= ~spin
        flags = np.where(mv, (flags & ~np.uint32(_capi.F_STATIC)) | np.uint32(_capi.F_HAS_VEL), flags).astype(np.uint32)
        for k in range(3):
            e["vel"][:, k] = np.where(mv, np.float32(60.0) * (uniform(SEED_MIX, idx, 1 + k) - np.float32(0.5)), 0).astype(np.float32)
    ext = h * scale[:, 0]                                        # world-space half extent
    if spinner_every:                                            # the 2-corner AABB of a box rotating about +y reaches h * sqrt(2) in x and z: keep it inside the section at every angle
        ext = np.where(spin, ext * np.float32(1.4143), ext).astype(np.float32)
    span = (np.float32(atomic) - np.float32(2.0) * ext) * np.float32(0.999)
    span = np.maximum(span, np.float32(0.0))
    a = np.float32(atomic)
    e["pos"][:, 0] = cx.astype(np.float32) * a + ext + span * u1
    e["pos"][:, 1] = cy.astype(np.float32) * a + ext + span * u2
    e["pos"][:, 2] = cz.astype(np.float32) * a + ext + span * u3
    for k in range(3):
        e["original"][:, 2 * k] = -h
        e["original"][:, 2 * k + 1] = h
    e["scale"] = scale
    e["flags"] = flags
    return e


def hopping_lattice(dims=(14, 14, 14), first_cell=121, atomic=64, every=2):
    """a lattice (one entity per level-0 world section, no shared sections) in which every `every`-th entity hops exactly one or two sections per
    unit of time along an axis: after a tick with dt = 1 no AABB straddles a section border, so a batch of movers touches unique world sections
    only (the batch the device-side re-bucket handles)"""
    ents = box_world(dims, first_cell=first_cell, atomic=atomic, mover_every=every)
    mv = (ents["flags"] & _capi.F_HAS_VEL) != 0
    idx = np.nonzero(mv)[0]
    a = np.float32(atomic)
    cell = np.floor(ents["pos"][idx] / a)
    ents["pos"][idx] = (cell + np.float32(0.5)) * a + (np.float32(6.0) * ((idx[:, None] * np.array([3, 5, 7])) % 5 - 2)).astype(np.float32)   # well inside the section
    ents["vel"][idx] = 0
    axis = idx % 3; sign = np.where((idx // 3) % 2 == 0, 1.0, -1.0).astype(np.float32)
    ents["vel"][idx, axis] = sign * a * np.where(idx % 5 == 0, 2.0, 1.0).astype(np.float32)
    return ents


def mixed_world(n, seed=1234, centre=(8192.0, 8192.0, 8192.0), spread=900.0, atomic=64):
    """Small adversarial world for parity tests: random sizes (unique sections at several levels
    and shared sections), static and active entities, spinners, movers, always-execute entities,
    several models / render systems / sortable buckets, clustered around `centre`."""
    idx = np.arange(n, dtype=np.uint64)
    e = np.zeros(n, ENTITY_DT)
    e["id"] = (idx * np.uint64(3) + np.uint64(7)).astype(np.uint32)          # non-dense ids
    u = lambda s: uniform(seed, idx, s)
    e["model_index"] = (u(0) * 5).astype(np.uint32)
    e["render_system"] = (u(1) * 2).astype(np.uint32)
    e["sortable"] = np.where(u(2) < 0.15, (u(3) * 4).astype(np.uint32), 0).astype(np.uint32)
    size_class = u(4)
    h = np.where(size_class < 0.6, 0.5 + 3.0 * u(5), np.where(size_class < 0.85, 8.0 + 30.0 * u(5), 40.0 + 150.0 * u(5))).astype(np.float32)
    hy = (h * (0.5 + u(6))).astype(np.float32); hz = (h * (0.5 + u(7))).astype(np.float32)
    e["original"][:, 0] = -h; e["original"][:, 1] = h; e["original"][:, 2] = -hy; e["original"][:, 3] = hy; e["original"][:, 4] = -hz; e["original"][:, 5] = hz
    for k in range(3):
        e["pos"][:, k] = np.float32(centre[k]) + np.float32(spread) * (np.float32(2.0) * u(8 + k) - np.float32(1.0))
    kind = u(11)
    flags = np.zeros(n, np.uint32)
    static = kind < 0.45
    flags[static] |= _capi.F_STATIC
    spin = (kind >= 0.45) & (kind < 0.7)
    flags[spin] |= _capi.F_HAS_ROT | _capi.F_HAS_ROTVEL
    accel_spin = spin & (u(12) < 0.3)
    flags[accel_spin] |= _capi.F_HAS_ROTACC
    mover = (kind >= 0.6) & (kind < 0.85)
    flags[mover] |= _capi.F_HAS_VEL
    accel_mv = mover & (u(13) < 0.4)
    flags[accel_mv] |= _capi.F_HAS_ACC
    scaled = u(14) < 0.3
    flags[scaled] |= _capi.F_HAS_SCALE
    rotated = (u(15) < 0.3) | spin
    flags[rotated] |= _capi.F_HAS_ROT
    flags[u(16) < 0.05] |= _capi.F_ALWAYS_EXEC
    flags[u(17) < 0.5] |= _capi.F_OOB_LOGIC
    for k in range(3):
        e["rot_axis"][:, k] = u(20 + k) - np.float32(0.5) + (np.float32(0.6) if k == 1 else 0)
        e["rotvel_axis"][:, k] = u(23 + k) - np.float32(0.5) + (np.float32(0.6) if k == 0 else 0)
        e["rotacc_axis"][:, k] = u(26 + k) - np.float32(0.5) + (np.float32(0.6) if k == 2 else 0)
        e["scale"][:, k] = np.float32(0.5) + np.float32(1.5) * u(29 + k)
        e["vel"][:, k] = np.float32(40.0) * (u(32 + k) - np.float32(0.5))
        e["acc"][:, k] = np.float32(10.0) * (u(35 + k) - np.float32(0.5))
    e["rot_angle"] = (np.float32(6.0) * (u(38) - np.float32(0.5))).astype(np.float32)
    e["rotvel"] = (np.float32(2.0) * (u(39) - np.float32(0.5))).astype(np.float32)
    e["rotacc"] = (np.float32(0.5) * (u(40) - np.float32(0.5))).astype(np.float32)
    e["vel"][u(41) < 0.1] = 0                                   # zero-velocity branch
    e["flags"] = flags
    return e


SEED_SCENE = 0x5EED0004
SAMPLE_SCENE_CAMERA = dict(position=(1000.0, 1000.0, 1150.0), direction=(0.0, 0.0, -1.0), far=1000.0)     # main.rs:24-33
SAMPLE_SCENE_WORLD = dict(outline_length=16384, atomic_length=64)   # threads/render_thread.rs:127, exports/load_models.rs:52 (release build)
SAMPLE_SCENE_MODEL_INDEX = {"yellowStar": 0, "blueStar": 1, "asteroid": 2, "wormhole": 3, "mine_producer": 4, "_user": 6}   # registration order; 5 = skyBox


def sample_scene(models, seed=SEED_SCENE, asteroids_per_sun=20):
    """configs[0]: the reference's own sample scene (`space_logic`), 45 entities.

    `models` maps model name -> OriginalAABB (6 floats), see tests/golden/sample_scene_models.json.
    Placement follows the reference's upload functions: stars solar_system/sun.rs:93-159, asteroids
    asteroid.rs:84-171 (its thread_rng draws replaced by the counter-based RNG, one stream per draw),
    wormhole wormhole.rs:62-73, mine producer mine_producer.rs:67-79, the user entity flows/pipeline.rs:125-151
    with the +-5 box of main.rs:35-41 at the camera position.  Entity ids in creation order: the user entity is
    created by ECS::new (objects/ecs.rs:141), then the instances in registration order (main.rs:59-63).
    All entities are non-static (EntityTransformationBuilder::new(.., false, ..)); stars sit in sortable bucket 1
    (write_sortable_component, sun.rs:123).  Lights are not modelled (out of the path)."""
    n = 1 + 2 + 2 * asteroids_per_sun + 2
    e = np.zeros(n, ENTITY_DT)
    e["id"] = np.arange(n, dtype=np.uint32)
    e["scale"][:] = 1.0
    deg = np.float32(np.pi / 180.0)
    ROT = _capi.F_HAS_ROT | _capi.F_HAS_ROTVEL | _capi.F_HAS_SCALE

    def put(i, model, pos, flags, scale=None, rot=None, rotvel=None, sortable=0):
        e["model_index"][i] = SAMPLE_SCENE_MODEL_INDEX[model]; e["sortable"][i] = sortable
        e["original"][i] = np.asarray(models[model], np.float32); e["pos"][i] = np.asarray(pos, np.float32); e["flags"][i] = flags
        if scale is not None: e["scale"][i] = np.float32(scale)
        if rot is not None: e["rot_axis"][i] = rot[0]; e["rot_angle"][i] = np.float32(rot[1])
        if rotvel is not None: e["rotvel_axis"][i] = rotvel[0]; e["rotvel"][i] = np.float32(rotvel[1])

    # user entity: OriginalAABB +-5, Position = camera, Velocity = Acceleration = 0 (pipeline.rs:125-144)
    put(0, "_user", SAMPLE_SCENE_CAMERA["position"], _capi.F_USER | _capi.F_HAS_VEL | _capi.F_HAS_ACC)
    # Rotation::default() = axis (1,0,0), angle 0 (movement_components.rs:41-47)
    put(1, "yellowStar", (950.0, 1000.0, 965.0), ROT, 10.0, ((1, 0, 0), 0.0), ((0, 1, 0), np.float32(-40.0) * deg), sortable=1)
    put(2, "blueStar", (1050.0, 1000.0, 965.0), ROT, 15.0, ((1, 0, 0), 0.0), ((0, 1, 0), np.float32(50.0) * deg), sortable=1)
    k = 3
    for sun in (1, 2):
        for j in range(asteroids_per_sun):
            i = np.uint64(k)
            xz = np.float32(360.0) * uniform(seed, i, 0); radius = np.float32(30.0) + np.float32(20.0) * uniform(seed, i, 1)
            y = np.float32(1000.0) + (np.float32(40.0) * uniform(seed, i, 2) - np.float32(20.0))
            rate = (np.float32(40.0) * uniform(seed, i, 3) - np.float32(20.0)) * deg
            x = np.float32(np.cos(xz * deg)) * radius + e["pos"][sun][0]                # calculate_position, asteroid.rs:173-179
            z = np.float32(np.sin(xz * deg)) * radius + e["pos"][sun][2]
            put(k, "asteroid", (x, y, z), ROT, 2.0, ((0, 1, 0), np.float32(0.1) * deg), ((0, 1, 0), rate))
            k += 1
    put(k, "wormhole", (970.0, 1000.0, 1000.0), _capi.F_HAS_SCALE, 5.0); k += 1
    put(k, "mine_producer", (980.0, 1000.0, 1000.0), ROT, 5.0, ((1, 0, 0), 0.0), ((1, 0, 0), np.float32(30.0) * deg)); k += 1
    assert k == n
    return e
