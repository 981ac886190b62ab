"""History / replay wire format: re_history_* against bytes written out by hand from bincode 1.3's layout rules and the
reference's type definitions (threads/public_common_structures.rs:7-16, objects/entity_change_request.rs:10-36, objects/ecs.rs:92-95,
exports/camera_object.rs:47-53, exports/movement_components.rs:6-39), the file pair of threads/history_thread.rs:150-205 /
helper_things/game_loader.rs:32-71, and (GPU) a recorded session replayed through Pipeline::debug_execute's loop."""
import struct

import numpy as np
import pytest

import render_engine_amd as R
from render_engine_amd import history as H

C = R._capi
IDS = dict(position=0xA1B2C3D4E5F60718, rotation=0x1111111111111111, scale=0x2222222222222222, velocity=0x3333333333333333,
           acceleration=0x4444444444444444, rotation_velocity=0x5555555555555555, rotation_acceleration=0x6666666666666666,
           has_moved=0x7777777777777777, has_rotated=0x8888888888888888)


def u32(v): return struct.pack("<I", v)
def u64(v): return struct.pack("<Q", v)
def f32s(*v): return struct.pack("<%df" % len(v), *v)
def seq3(v): return u64(3) + f32s(*v)            # nalgebra 0.25 TVec3<f32> through serde: a sequence of three elements


def changes(rows):
    ch = np.zeros(len(rows), R.CHANGE_DT)
    for k, r in enumerate(rows):
        ch[k] = r
    return ch


def test_frame_change_records_byte_exact():
    h = H.History(IDS)
    h.camera_view_change((1.5, -2.0, 3.25), (0.0, 0.0, -1.0))
    h.camera_stationary()
    h.delta_time(0.016)
    h.draw_distances_change(0.1, 1000.0, 45.0)
    h.window_dimensions_change(1280, 720)
    h.end_frame()
    want = [u32(0) + seq3((1.5, -2.0, 3.25)) + seq3((0.0, 0.0, -1.0)),          # CameraViewChange(SerializableCameraInfo { position, direction })
            u32(1),                                                              # CameraStationary
            u32(2) + f32s(0.016),                                                # DeltaTime(f32)
            u32(3) + f32s(0.1, 1000.0, 45.0),                                    # DrawDistancesChange(f32, f32, f32)
            u32(4) + struct.pack("<ii", 1280, 720),                              # WindowDimensionsChange((i32, i32))
            u32(6)]                                                              # EndFrameChange
    assert len(h) == len(want)
    for k, w in enumerate(want):
        assert h.encode(k) == w, k
    a = H.History(IDS, vec3_as_array=True)
    a.camera_view_change((1.5, -2.0, 3.25), (0.0, 0.0, -1.0))
    assert a.encode(0) == u32(0) + f32s(1.5, -2.0, 3.25) + f32s(0.0, 0.0, -1.0)
    h.close(); a.close()


def test_entity_change_record_byte_exact():
    h = H.History(IDS)
    h.entity_change(changes([(C.CHANGE_MODIFY, 7, C.C_POSITION, 0, (10.0, 20.0, 30.0, 0.0)),
                             (C.CHANGE_MODIFY, 8, C.C_ROTATION_VEL, 0, (0.0, 1.0, 0.0, 0.5)),
                             (C.CHANGE_DELETE, 9, 0, 0, (0, 0, 0, 0)),
                             (C.CHANGE_MAKE_STATIC, 3, 0, 0, (0, 0, 0, 0)),
                             (C.CHANGE_WAKE_UP, 4, 0, 0, (0, 0, 0, 0)),
                             (C.CHANGE_REMOVE_COMPONENT, 5, C.C_VELOCITY, 0, (0, 0, 0, 0))]))
    want = (u32(5) + u64(6)                                                                                  # EntityChange(Vec<EntityChangeInformation>), 6 elements
            # ModifyRequest(EntityChangeRequest { entity_id: EntityId(7), type_id: vec![(TypeIdentifier { t: [position] }, 12 bytes of Position)] })
            + u32(5) + u32(7) + u64(1) + u64(IDS["position"]) + u64(12) + f32s(10.0, 20.0, 30.0)
            # ... VelocityRotation { axis: vec3, rotation: f32 } = 16 bytes
            + u32(5) + u32(8) + u64(1) + u64(IDS["rotation_velocity"]) + u64(16) + f32s(0.0, 1.0, 0.0, 0.5)
            + u32(9) + u32(9)                                                                                 # DeleteRequest(EntityId(9))
            + u32(10) + u32(3)                                                                                # MakeObjectStatic(EntityId(3))
            + u32(11) + u32(4)                                                                                # WakeUpRequest(EntityId(4))
            + u32(6) + u32(5) + u64(IDS["velocity"]))                                                         # RemoveComponent((EntityId(5), TypeIdentifier))
    assert h.encode(0) == want
    h.close()


def test_file_pair_round_trip_and_lookup_file(tmp_path):
    rng = np.random.default_rng(3)
    h = H.History(IDS)
    h.set_state(b"ECS-BLOB-\x00\x01\x02", b"TREE" * 5)
    recorded = []
    for f in range(40):
        h.delta_time(0.01 + 0.001 * f); recorded.append((C.FC["DELTA_TIME"],))
        if f % 3:
            pos, d = rng.normal(size=3).astype(np.float32), rng.normal(size=3).astype(np.float32)
            h.camera_view_change(pos, d); recorded.append((C.FC["CAMERA_VIEW_CHANGE"], pos, d))
        else:
            h.camera_stationary(); recorded.append((C.FC["CAMERA_STATIONARY"],))
        if f % 4 == 1:
            ch = changes([(C.CHANGE_MODIFY, int(rng.integers(1, 100)), int(rng.integers(0, 7)), 0, tuple(rng.normal(size=4).astype(np.float32))) for _ in range(int(rng.integers(1, 9)))] +
                         [(C.CHANGE_DELETE, 5 + f, 0, 0, (0, 0, 0, 0)), (C.CHANGE_REMOVE_COMPONENT, 6 + f, C.C_SCALE, 0, (0, 0, 0, 0))])
            for c in ch:                                       # 3-float components carry no 4th value on the wire
                if c["kind"] == C.CHANGE_MODIFY and c["component"] in (C.C_POSITION, C.C_SCALE, C.C_VELOCITY, C.C_ACCELERATION):
                    c["value"][3] = 0
            h.entity_change(ch); recorded.append((C.FC["ENTITY_CHANGE"], ch))
        if f == 7:
            h.draw_distances_change(0.5, 2000.0, 60.0); recorded.append((C.FC["DRAW_DISTANCES_CHANGE"],))
            h.window_dimensions_change(1920, 1080); recorded.append((C.FC["WINDOW_DIMENSIONS_CHANGE"],))
        h.end_frame(); recorded.append((C.FC["END_FRAME_CHANGE"],))
    hp, lp = tmp_path / "gameplay_history.txt", tmp_path / "gameplay_byte_lookup.txt"
    h.write(hp, lp)
    # the lookup file: one decimal length per line, every line (also the last) ends in '\n' (history_thread.rs:200-205); the lengths tile the history file
    text = lp.read_text()
    assert text.endswith("\n")
    lens = [int(x) for x in text.split("\n")[:-1]]
    blob = hp.read_bytes()
    assert sum(lens) == len(blob) and len(lens) == 2 + len(h)
    assert blob[:lens[0]] == b"ECS-BLOB-\x00\x01\x02" and blob[lens[0]:lens[0] + lens[1]] == b"TREE" * 5
    off = lens[0] + lens[1]
    for k in range(len(h)):
        assert blob[off:off + lens[2 + k]] == h.encode(k)
        off += lens[2 + k]
    g = H.History.load(hp, lp, IDS)
    assert g.state() == (b"ECS-BLOB-\x00\x01\x02", b"TREE" * 5)
    assert len(g) == len(h) == len(recorded)
    for k, rec in enumerate(recorded):
        kind, f, i, ch = g.get(k)
        assert kind == rec[0]
        assert g.encode(k) == h.encode(k)
        if kind == C.FC["CAMERA_VIEW_CHANGE"]:
            np.testing.assert_array_equal(f[:3], rec[1]); np.testing.assert_array_equal(f[3:], rec[2])
        if kind == C.FC["ENTITY_CHANGE"]:
            assert ch.tobytes() == rec[1].tobytes()
    assert g.frame_indexes() == h.frame_indexes() and len(g.frame_indexes()) == 40
    h.close(); g.close()


def test_requests_outside_the_path_and_malformed_files(tmp_path):
    hp, lp = tmp_path / "h.bin", tmp_path / "l.txt"
    # a ModifyRequest that carries two components (one of them a HasMoved marker, which re_tick maintains itself) decodes to the component this path carries
    rec = u32(5) + u64(1) + u32(5) + u32(12) + u64(2) + u64(IDS["has_moved"]) + u64(0) + u64(IDS["scale"]) + u64(12) + f32s(2.0, 2.0, 2.0)
    hp.write_bytes(rec); lp.write_text("0\n0\n%d\n" % len(rec))
    g = H.History.load(hp, lp, IDS)
    kind, _, _, ch = g.get(0)
    assert kind == C.FC["ENTITY_CHANGE"] and len(ch) == 1 and ch[0]["entity_id"] == 12 and ch[0]["component"] == C.C_SCALE and tuple(ch[0]["value"][:3]) == (2.0, 2.0, 2.0)
    g.close()
    # AddSortableComponent(EntityId, TypeIdentifier) is outside this path: refused with a message, not skipped
    rec = u32(5) + u64(1) + u32(3) + u32(12) + u64(99)
    hp.write_bytes(rec); lp.write_text("0\n0\n%d\n" % len(rec))
    with pytest.raises(R.RenderEngineError, match="outside this path"):
        H.History.load(hp, lp, IDS)
    # lengths that do not tile the file, a truncated record, an unknown variant
    hp.write_bytes(u32(1)); lp.write_text("0\n0\n9\n")
    with pytest.raises(R.RenderEngineError, match="does not match"):
        H.History.load(hp, lp, IDS)
    hp.write_bytes(u32(2) + b"\x00\x00"); lp.write_text("0\n0\n6\n")
    with pytest.raises(R.RenderEngineError, match="wrong length"):
        H.History.load(hp, lp, IDS)
    hp.write_bytes(u32(17)); lp.write_text("0\n0\n4\n")
    with pytest.raises(R.RenderEngineError, match="unknown FrameChange"):
        H.History.load(hp, lp, IDS)
    with pytest.raises(R.RenderEngineError, match="cannot open"):
        H.History.load(tmp_path / "missing", lp, IDS)
    # recording a change this format cannot carry is refused at record time
    h = H.History(IDS)
    with pytest.raises(R.RenderEngineError, match="TypeIdentifier"):
        h.entity_change(changes([(C.CHANGE_MODIFY, 1, C.C_TRANSFORMATION, 0, (0, 0, 0, 0))]))
    assert len(h) == 0
    h.close()


@pytest.mark.gpu
def test_recorded_session_replays_bit_exact(tmp_path):
    """a session driven directly (GPU pipeline and oracle side by side, every frame compared), recorded as the history thread would,
    written to the file pair, loaded, and replayed on a fresh pipeline and a fresh oracle world by debug_execute's loop: every replayed
    frame equals the frame of the live session"""
    import oracle as ro
    from helpers import to_oracle, oracle_camera, assert_render_equal
    from test_gpu_parity import random_changes
    ents = R.synthetic.mixed_world(2500, seed=33, spread=500.0)
    live, w = R.Pipeline(16384, 64), ro.World(16384, 64)
    assert live.register_model_instances(ents) == w.register(to_oracle(ents))
    rng = np.random.default_rng(8)
    h = H.History(IDS)
    cam = H.ReplayCamera((8192.0, 8192.0, 8600.0), (0.0, 0.0, -1.0), 900.0)
    start = (cam.position.copy(), cam.direction.copy())
    frames = []
    for f in range(10):
        c = cam.camera(); oc = oracle_camera(c)
        w.cull(oc)
        g = live.cull_and_pack(c)
        assert_render_equal(g, w.render(oc))
        frames.append(g)
        dt = 0.016 + 0.002 * f
        h.delta_time(dt)
        if f % 3 == 2:
            h.camera_stationary()
        else:
            cam.position = (cam.position + np.float32([25.0, -6.0, -30.0])).astype(np.float32)
            cam.direction = np.float32([0.04 * f, 0.0, -1.0])
            h.camera_view_change(cam.position, cam.direction)
        live.tick(dt); w.tick(oc, dt)
        if f % 2 == 0:
            ch = random_changes(R, ents, rng, 60, set())
            live.apply_changes(ch); w.apply_changes(ch.view(ro.CHANGE_DT))
            h.entity_change(ch)
        if f == 4:
            cam.near, cam.far, cam.fov = 0.2, 1400.0, 50.0; h.draw_distances_change(0.2, 1400.0, 50.0)
        if f == 6:
            cam.window = (1920, 1080); h.window_dimensions_change(1920, 1080)
        h.end_frame()
    live.close(); w.close()
    hp, lp = tmp_path / "gameplay_history.txt", tmp_path / "gameplay_byte_lookup.txt"
    h.write(hp, lp); h.close()
    loaded = H.History.load(hp, lp, IDS)

    class OracleWorld:                                  # the frame calls of Pipeline over the oracle, for the same replay loop
        def __init__(self):
            self.w = ro.World(16384, 64); self.w.register(to_oracle(ents)); self.oc = None
        def cull_and_pack(self, c):
            self.oc = oracle_camera(c); self.w.cull(self.oc); return self.w.render(self.oc)
        def tick(self, dt): self.w.tick(self.oc, dt)
        def apply_changes(self, ch): self.w.apply_changes(ch.view(ro.CHANGE_DT))

    p = R.Pipeline(16384, 64); p.register_model_instances(ents)
    got = H.replay(p, loaded, H.ReplayCamera(start[0], start[1], 900.0))
    ow = OracleWorld()
    want = H.replay(ow, loaded, H.ReplayCamera(start[0], start[1], 900.0))
    assert len(got) == len(want) == len(frames) == 10
    for k in range(10):
        assert_render_equal(got[k], want[k])
        assert_render_equal(got[k], frames[k])
    st = p.stats(); assert st["n_seal_waits"] == 0 and st["n_sync_fallbacks"] == 0
    p.close(); ow.w.close(); loaded.close()
