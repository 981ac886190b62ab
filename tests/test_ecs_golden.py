"""The reference's own ECS known answers (objects/ecs.rs #[cfg(test)], transcribed as data into tests/golden/ecs_cases.json) replayed
through the C ABI: presence bitset (re_ecs_bitset), getters (re_read_component), remove_component / remove_entity (re_apply_changes:
RE_CHANGE_REMOVE_COMPONENT / RE_CHANGE_DELETE), write_component (RE_CHANGE_MODIFY) and the has-components query (re_ecs_query ==
ECS::get_indexes_for_components).

The reference's tests register three test-local component types; the engine's ECS has a fixed registration (flows/logic_flow.rs:83-110).
The replay maps them onto three engine components that can be written and removed after registration:
    Position -> Rotation, Velocity -> Velocity, Acceleration -> Acceleration      (write / remove cases; entities start blank)
    Position -> Rotation, Velocity -> Scale                                       (query cases; components present from registration)
and translates every expected bitset byte through the two registration orders (bit k of the reference byte <-> the engine bit of the
mapped component).  Only the three mapped bits are compared: the engine's entities also carry Position, TransformationMatrix, ModelId, ...
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "ecs_cases.json")))


@pytest.fixture(scope="module")
def R():
    import render_engine_amd as R
    return R


def blank_entities(R, n, with_flags=None):
    """n entities far apart; every one owns a slot in the dynamic table (uploaded with Velocity, which the replay removes first) so that
    Velocity / Acceleration can be written later"""
    C = R._capi
    e = np.zeros(n, R.ENTITY_DT)
    e["rot_axis"][:, 0] = 1; e["rotvel_axis"][:, 0] = 1; e["rotacc_axis"][:, 0] = 1; e["scale"][:] = 1
    e["id"] = np.arange(n); e["original"][:] = (-1, 1, -1, 1, -1, 1)
    e["pos"][:, 0] = 8000 + 100 * np.arange(n); e["pos"][:, 1] = 8000; e["pos"][:, 2] = 8000
    e["flags"] = C.F_HAS_VEL if with_flags is None else with_flags
    return e


def value_for(R, comp):
    C = R._capi
    return {C.C_ROTATION: (0.0, 1.0, 0.0, 0.5), C.C_SCALE: (2.0, 2.0, 2.0, 0.0), C.C_VELOCITY: (1.0, 2.0, 3.0, 0.0), C.C_ACCELERATION: (0.5, 0.0, 0.0, 0.0)}[comp]


@pytest.mark.parametrize("case", [c for c in CASES["cases"] if "initial" not in c], ids=lambda c: c["name"])
def test_write_remove_cases(R, case):
    C = R._capi
    comp = {"Position": C.C_ROTATION, "Velocity": C.C_VELOCITY, "Acceleration": C.C_ACCELERATION}
    ebit = {"Position": C.ECS_BIT["ROTATION"], "Velocity": C.ECS_BIT["VELOCITY"], "Acceleration": C.ECS_BIT["ACCELERATION"]}
    rbit = CASES["registration_order"]
    n = case["entities"]
    p = R.Pipeline(16384, 64)
    p.register_model_instances(blank_entities(R, n))
    p.cull_and_pack(R.Camera((8000, 8000, 8300), (0, 0, -1), 1000.0))       # apply_change runs inside a frame
    ch = np.zeros(n, R.CHANGE_DT)
    for i in range(n):
        ch[i] = (C.CHANGE_REMOVE_COMPONENT, i, C.C_VELOCITY, 0, (0, 0, 0, 0))     # reach the reference's freshly created entity: none of the three
    p.apply_changes(ch)

    def change(kind, ent, name=None):
        c1 = np.zeros(1, R.CHANGE_DT)
        c1[0] = (kind, ent, comp[name] if name else 0, 0, value_for(R, comp[name]) if (name and kind == C.CHANGE_MODIFY) else (0, 0, 0, 0))
        p.apply_changes(c1)

    def mapped_bits(ent):
        b = p.ecs_bitset(ent)
        return sum(1 << rbit[nm] for nm in comp if (b >> ebit[nm]) & 1)

    for st in case["steps"]:
        (op, arg), = st.items()
        if op == "write":
            change(C.CHANGE_MODIFY, arg[0], arg[1])
        elif op == "remove":
            change(C.CHANGE_REMOVE_COMPONENT, arg[0], arg[1])
        elif op == "remove_entity":
            change(C.CHANGE_DELETE, arg)
        elif op == "expect_bitset":
            assert mapped_bits(arg[0]) == arg[1], (case["name"], st)
        elif op == "expect_some":
            for nm in arg[1]:
                got = p.read_component(arg[0], comp[nm])                          # check_getters_some: the value written
                want = np.asarray(value_for(R, comp[nm])[:len(got)], np.float32)
                if comp[nm] == C.C_ROTATION:
                    want[:3] /= np.linalg.norm(want[:3])
                np.testing.assert_allclose(got, want, rtol=1e-6)
        elif op == "expect_none":
            for nm in arg[1]:
                with pytest.raises(R.RenderEngineError):                          # check_getters_none: get_copy -> None
                    p.read_component(arg[0], comp[nm])
        elif op == "expect_written":
            assert p.has_component(arg[0], comp[arg[1]]) == arg[2]
        else:
            raise AssertionError(op)
    p.close()


@pytest.mark.parametrize("case", [c for c in CASES["cases"] if "initial" in c], ids=lambda c: c["name"])
def test_query_cases(R, case):
    C = R._capi
    comp = {"Position": C.C_ROTATION, "Velocity": C.C_SCALE}
    flag = {"Position": C.F_HAS_ROT, "Velocity": C.F_HAS_SCALE}
    n = case["entities"]
    fl = np.zeros(n, np.uint32)
    for nm, ids in case["initial"].items():
        fl[ids] |= flag[nm]
    e = blank_entities(R, n, with_flags=fl)
    e["rot_axis"][:] = (0, 1, 0); e["rot_angle"] = 0.25; e["scale"][:] = 2
    p = R.Pipeline(16384, 64)
    p.register_model_instances(e)
    for st in case["steps"]:
        names, want = st["query"]
        got = p.get_indexes_for_components([comp[nm] for nm in names])
        assert list(got) == want, (case["name"], names)
    # a removed component drops out of the query, a removed entity out of every query (remove_component / remove_entity)
    p.cull_and_pack(R.Camera((8000, 8000, 8300), (0, 0, -1), 1000.0))
    ch = np.zeros(2, R.CHANGE_DT)
    first = case["initial"]["Velocity"][0]
    ch[0] = (C.CHANGE_REMOVE_COMPONENT, first, C.C_SCALE, 0, (0, 0, 0, 0)); ch[1] = (C.CHANGE_DELETE, case["initial"]["Velocity"][-1], 0, 0, (0, 0, 0, 0))
    p.apply_changes(ch)
    assert list(p.get_indexes_for_components([C.C_SCALE])) == case["initial"]["Velocity"][1:-1]
    assert len(p.get_indexes_for_components([])) == n - 1
    p.close()
