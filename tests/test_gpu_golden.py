"""The reference's own known-answer tests for the spatial hash (tests/golden/tree_cells.json, transcribed from
world/bounding_box_tree_v2.rs #[cfg(test)]) replayed on the HIP path through the C ABI: section assignment of single boxes, and
the add / remove sequences (removal == DeleteRequest through re_apply_changes, which patches the resident section table)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tree_cells.json")))


@pytest.fixture(scope="module")
def R():
    import render_engine_amd as R
    return R


def entities(R, boxes):
    e = np.zeros(len(boxes), R.ENTITY_DT)
    e["id"] = np.arange(len(boxes), dtype=np.uint32)
    e["scale"][:] = 1.0
    for i, b in enumerate(boxes):
        e["original"][i] = np.asarray(b, np.float32)          # identity transform: StaticAABB == OriginalAABB
    return e


def unpack(k):
    k = int(k)
    return ((k >> 48) & 0xFFFF, (k >> 32) & 0xFFFF, (k >> 16) & 0xFFFF, k & 0xFFFF)


@pytest.mark.parametrize("case", G["assign"], ids=lambda c: c["ref"])
def test_cell_assignment_known_answers_on_gpu(R, case):
    p = R.Pipeline(G["outline"], G["atomic"])
    assert p.register_model_instances(entities(R, [case["aabb"]])) == 0
    s = p.sections()
    assert sorted(unpack(k) for k in s["keys"]) == sorted(tuple(i) for i in case["ids"])
    st = p.stats()
    assert st["n_shared_sections"] == (1 if case["kind"] == "shared" else 0)
    if case["kind"] == "unique":
        assert int(s["n_local"][0]) == 1
    else:
        assert int(s["n_local"].sum()) == 0                   # the entity lives in the shared section, its linking sections are empty
    p.close()


def check_state(R, p, exp):
    s = p.sections()
    got = {unpack(k): int(n) for k, n in zip(s["keys"], s["n_local"])}
    want = {tuple(int(v) for v in k.split(",")): len(c["local"]) for k, c in exp["cells"].items()}
    assert got == want
    assert p.stats()["n_shared_sections"] == len(exp["shared"])


@pytest.mark.parametrize("seq", G["sequences"], ids=lambda s: s["name"])
def test_tree_sequences_known_answers_on_gpu(R, seq):
    ops = seq["ops"]
    boxes = [op[1] for op in ops if op[0] == "add"]
    n_adds_before_first_other = next(i for i, op in enumerate(ops) if op[0] != "add")
    assert n_adds_before_first_other == len(boxes)            # every sequence adds first, then removes / expects
    p = R.Pipeline(G["outline"], G["atomic"])
    assert p.register_model_instances(entities(R, boxes)) == 0
    p.cull_and_pack(R.Camera((128, 128, 300), (0, 0, -1), 500.0))     # apply_change runs inside a frame
    for op in ops[len(boxes):]:
        if op[0] == "remove":
            ch = np.zeros(1, R.CHANGE_DT); ch[0] = (R._capi.CHANGE_DELETE, op[1], 0, 0, (0, 0, 0, 0))
            p.apply_changes(ch)
        else:
            check_state(R, p, op[1])
    p.close()
