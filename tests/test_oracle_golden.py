"""Pins the CPU oracle against the reference's own known-answer tests
(reference: src/world/bounding_box_tree_v2.rs #[cfg(test)], transcribed as data in
tests/golden/tree_cells.json)."""
import json
import os

import numpy as np
import pytest

import oracle as ro

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tree_cells.json")))


def key_of(i):
    return ro.pack_key(*i)


@pytest.mark.parametrize("case", G["assign"], ids=lambda c: c["ref"].split(" ")[0] + str(c["aabb"][:2]))
def test_cell_assignment_known_answers(case):
    L = ro.lib()
    keys = (ro.C.c_uint64 * 8)(); oob = ro.C.c_int()
    n = L.ro_assign_cells(ro.aabb(case["aabb"]), G["outline"], G["atomic"], keys, ro.C.byref(oob))
    assert oob.value == 0
    got = [ro.unpack_key(keys[i]) for i in range(n)]
    assert got == [tuple(i) for i in case["ids"]]
    assert (n == 1) == (case["kind"] == "unique")


def test_to_aabb_known_answer():
    t = G["to_aabb"]
    a = ro.lib().ro_key_to_aabb(key_of(t["id"]), t["atomic"])
    assert [a.xmin, a.xmax] == t["x"] and [a.zmin, a.zmax] == t["z"] and [a.ymin, a.ymax] == t["y"]


def test_max_level():
    L = ro.lib()
    assert L.ro_max_level(256, 32) == 3          # bounding_box_tree_v2.rs:1475 fixtures
    assert L.ro_max_level(16384, 64) == 8        # render_thread.rs:127 / load_models.rs:52
    assert L.ro_max_level(16384, 128) == 7       # debug build (main.rs:48-51)
    assert L.ro_max_level(16384, 32) == 9


def check_state(w, exp):
    cells = w.cells()
    got_keys = [ro.unpack_key(k) for k in cells["keys"]]
    assert sorted(got_keys) == sorted(tuple(int(v) for v in k.split(",")) for k in exp["cells"])
    shared = w.shared_sections()
    assert sorted(tuple(map(ro.unpack_key, s["keys"])) for s in shared) == sorted(tuple(tuple(i) for i in s["ids"]) for s in exp["shared"])
    for s in exp["shared"]:
        m = [x for x in shared if [ro.unpack_key(k) for k in x["keys"]] == [tuple(i) for i in s["ids"]]][0]
        assert sorted(m["active"].tolist()) == sorted(s["entities"])
    for kstr, c in exp["cells"].items():
        key = key_of([int(v) for v in kstr.split(",")])
        local, static = w.cell_entities(key)
        assert sorted(local.tolist()) == sorted(c["local"]) and len(static) == 0
        idx = list(cells["keys"]).index(key)
        assert cells["n_shared"][idx] == len(c["shared"])
    for eid, (kind, ids) in exp["lookup"].items():
        k, keys = w.lookup(int(eid))
        if kind == "unique":
            assert k == 1 and ro.unpack_key(keys[0]) == tuple(ids)
        else:
            assert k == 2 and [ro.unpack_key(x) for x in keys] == [tuple(i) for i in ids]
    for kstr, rel in exp.get("related", {}).items():               # related_world_sections (check_related_world_sections :1527-1543)
        assert w.related_sections(key_of([int(v) for v in kstr.split(",")])) == sorted(key_of(r) for r in rel)
    # every other entity has no lookup entry
    for eid in range(8):
        if str(eid) not in exp["lookup"]:
            assert w.lookup(eid)[0] == 0


@pytest.mark.parametrize("seq", G["sequences"], ids=lambda s: s["name"])
def test_tree_sequences_known_answers(seq):
    w = ro.World(G["outline"], G["atomic"])
    next_id = 0
    for op in seq["ops"]:
        if op[0] == "add":
            assert w.tree_add(next_id, op[1]) == 0
            next_id += 1
        elif op[0] == "remove":
            w.tree_remove(op[1])
        else:
            check_state(w, op[1])
    w.close()


def test_find_related_entities_known_answer():
    """find_related_entities (:2220-2303): from any section of the relationship the search reaches every section of it -- through
    the parent (2,0,0,0) also the sibling (1,1,0,0), which exists only as a link of the shared section -- and the shared section
    once; the unrelated section finds itself alone"""
    F = G["find_related"]
    w = ro.World(G["outline"], G["atomic"])
    for i, box in enumerate(F["adds"]):
        assert w.tree_add(i, box) == 0
    w.end_of_changes()
    for q in F["queries"]:
        for start in q["from"]:
            uniq, shared = w.find_related(key_of(start))
            assert sorted(uniq) == sorted(key_of([int(v) for v in k.split(",")]) for k in q["unique"])
            for k, ents in q["unique"].items():
                local, static = w.cell_entities(key_of([int(v) for v in k.split(",")]))
                assert sorted(local.tolist()) == ents
            assert sorted(shared) == sorted(tuple(key_of(i) for i in s["ids"]) for s in q["shared"])
    w.close()
