"""Shared helpers of the parity tests: scene conversion and result comparison."""
import numpy as np

import oracle as ro


def to_oracle(ents):
    """render_engine_amd.ENTITY_DT and oracle.ENTITY_DT share one byte layout."""
    assert ents.dtype.itemsize == ro.ENTITY_DT.itemsize
    return np.ascontiguousarray(ents).view(ro.ENTITY_DT)


def oracle_camera(cam):
    """oracle Camera struct from a render_engine_amd.Camera (same P*V bits on both sides)."""
    return ro.make_camera(cam.position, cam.direction, cam.far_draw_distance, lod=cam.level_of_views, pv=cam.projection_view)


def groups_as_dict(res):
    """{(model_index, render_system, sortable): sorted ids} and id -> matrix"""
    out = {}
    for g in res["groups"]:
        b, c = int(g["begin"]), int(g["count"])
        out[(int(g["model_index"]), int(g["render_system"]), int(g["sortable"]))] = np.sort(res["ids"][b:b + c])
    return out


def assert_render_equal(gpu, cpu):
    assert gpu["total"] == cpu["total"], (gpu["total"], cpu["total"])
    gg, cg = groups_as_dict(gpu), groups_as_dict(cpu)
    assert set(gg) == set(cg), (sorted(set(gg) ^ set(cg)))
    for k in cg:
        np.testing.assert_array_equal(gg[k], cg[k], err_msg=f"group {k}")
    # group ranges tile [0,total) back to back
    gr = np.sort(gpu["groups"], order="begin")
    assert int(gr["begin"][0]) == 0 if len(gr) else True
    assert np.all(gr["begin"][1:] == gr["begin"][:-1] + gr["count"][:-1])
    # matrices: bit-exact per instance.  An entity can appear more than once with different bytes (a ghost of the frozen static cache
    # next to the live entity), so instances are ordered by (id, matrix bits) on both sides.
    def order(ids, mats):
        bits = np.ascontiguousarray(mats, np.float32).view(np.uint32).reshape(len(ids), 16)
        return np.lexsort(tuple(bits[:, k] for k in range(15, -1, -1)) + (ids,))
    og, oc = order(gpu["ids"], gpu["mats"]), order(cpu["ids"], cpu["mats"])
    np.testing.assert_array_equal(gpu["ids"][og], cpu["ids"][oc])
    np.testing.assert_array_equal(gpu["mats"][og], cpu["mats"][oc])


def expand_vis(keys, mult):
    return np.sort(np.repeat(keys, mult.astype(np.int64)))
