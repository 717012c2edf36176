"""The CPU oracle against everything that can pin it without a GPU: the reference's own fixtures (the lossy
course JPEGs, coarse), the grid-occupancy figures derived independently from the scene files, hand-computed
primitive known answers, and the committed golden vectors (bit-exact, guards the oracle against drift)."""
import os

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN, scene_path
from oracle.ref_oracle import RefScene, quantize

FLT_MAX = np.finfo(np.float32).max


# --- the reference's own fixtures: course renders of scenes 1-3 (JPEG, different renderer) ---------------------------
def _with_camera(tmp_path, name, which):
    """The scene file with its `which`-th camera line active (the files list their alternative cameras as comments:
    res/scene1.test:9-12, res/scene2.test:17-19) and every other one commented out."""
    out, k = [], 0
    for line in open(scene_path(name)).read().splitlines():
        if line.lstrip("#").startswith("camera "):
            k += 1
            line = line.lstrip("#") if k == which else "#" + line.lstrip("#")
        out.append(line)
    path = tmp_path / f"{name}-camera{which}.test"
    path.write_text("\n".join(out) + "\n")
    return str(path)


# Measured agreement of the oracle with each course JPEG (lossy, rendered by the course's own renderer), printed by this
# test with -s: mean absolute error in 8-bit steps over the whole image, overlap (IoU) of the silhouettes, and the
# largest difference of the mean colour over the interior of the common silhouette (eroded by four pixels, away from
# the JPEG's ringing at edges):
#   scene1 camera1 0.503 0.9912 0.37   camera2 0.574 0.9873 0.38   camera3 0.258 0.9664 0.12   camera4 0.302 0.9751 0.34
#   scene2 camera1 1.004 0.9857 0.53   camera2 1.001 0.9897 0.47   camera3 1.077 0.9889 0.30   scene3 1.253 0.9837 1.17
# The bounds below are those figures plus a small margin.  scene1 is the only shaded fixture (a quad lit by a point and
# a directional light, diffuse only): its interior mean colour pins the Lambert term and both light kinds to within
# half an 8-bit step; scenes 2 and 3 are ambient-only and pin camera, transform stack and both primitive tests.
@pytest.mark.parametrize("name,which,jpg,max_mae,min_iou,max_mean_diff", [
    ("scene1", 1, "scene1-camera1.jpg", 0.65, 0.986, 0.6), ("scene1", 2, "scene1-camera2.jpg", 0.72, 0.982, 0.6),
    ("scene1", 3, "scene1-camera3.jpg", 0.40, 0.961, 0.6), ("scene1", 4, "scene1-camera4.jpg", 0.45, 0.970, 0.6),
    ("scene2", 1, "scene2-camera1.jpg", 1.15, 0.980, 0.8), ("scene2", 2, "scene2-camera2.jpg", 1.15, 0.984, 0.8),
    ("scene2", 3, "scene2-camera3.jpg", 1.25, 0.984, 0.8), ("scene3", 1, "scene3.jpg", 1.40, 0.978, 1.5)])
def test_oracle_matches_course_render(tmp_path, name, which, jpg, max_mae, min_iou, max_mean_diff):
    from scipy.ndimage import binary_erosion
    s = RefScene(_with_camera(tmp_path, name, which))
    rgb, ids, _, _ = s.render()
    want = np.asarray(Image.open(os.path.join(GOLDEN, jpg)).convert("RGB")).astype(int)
    got = quantize(rgb).astype(int)
    assert got.shape == want.shape == (480, 640, 3)
    mae = np.abs(got - want).mean()
    sil_got, sil_want = ids >= 0, want.sum(axis=2) > 24
    iou = (sil_got & sil_want).sum() / (sil_got | sil_want).sum()
    inner = binary_erosion(sil_got & sil_want, iterations=4)
    mean_diff = np.abs(got[inner].mean(axis=0) - want[inner].mean(axis=0)).max()
    print(f"{name} camera{which}: mean abs error {mae:.3f}, silhouette IoU {iou:.4f}, interior mean colour differs by {mean_diff:.2f}")
    assert mae < max_mae and iou > min_iou
    assert inner.sum() > 20000 and mean_diff < max_mean_diff
    if name == "scene1":
        # the lit quad: 99 % of its interior pixels within a few 8-bit steps of the course render, channel by channel
        assert np.percentile(np.abs(got - want)[inner], 99) <= 8


# --- grid occupancy figures of the 15^3 reference grid, derived from the scene files alone --------------------------
@pytest.mark.parametrize("name,occupied,refs,max_cell", [
    ("scene1", 225, 450, 2), ("scene4-specular", 1936, 14228, 23), ("scene5", 3375, 426075, 550),
    ("scene6", 2961, 16128, 38), ("scene7", 1833, 2122824, 12702)])
def test_reference_grid_occupancy(name, occupied, refs, max_cell):
    s = RefScene(scene_path(name))
    _, start, ids = s.grid()
    count = np.diff(start)
    assert len(ids) == refs
    assert (count > 0).sum() == occupied
    assert count.max() == max_cell
    # lists are in file order
    for c in np.nonzero(count > 1)[0][:200]:
        seg = ids[start[c]:start[c + 1]]
        assert np.all(np.diff(seg) > 0)


def test_box_maximum_is_seeded_with_smallest_positive_float():
    # a primitive entirely at negative coordinates still reports a box maximum of ~0 (src/geometries.h:146-152)
    s = RefScene(scene_path("scene7"))
    box = s.prims()["box"]
    assert (box[:, 3:] > 0).all()
    assert np.isclose(box[:, 3:].min(), np.finfo(np.float32).tiny)


# --- primitive known answers (hand-computed) --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def kat_scene(tmp_path_factory):
    p = tmp_path_factory.mktemp("kat") / "kat.test"
    p.write_text("size 8 8\ncamera 0 0 5 0 0 0 0 1 0 45\n"
                 "vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\n"
                 "sphere 0 0 0 1\n"      # primitive 0
                 "tri 0 1 2\n")          # primitive 1
    return RefScene(str(p), use_grid=False)


def test_sphere_known_answers(kat_scene):
    h = kat_scene.prim_hit(0, [0, 0, 5], [0, 0, -1])          # two roots 4 and 6
    assert h["ok"] and h["t"] == 4.0 and np.array_equal(h["n"], [0, 0, 1])
    h = kat_scene.prim_hit(0, [1, 0, 5], [0, 0, -1])          # tangent: discriminant exactly 0
    assert h["ok"] and h["t"] == 5.0
    h = kat_scene.prim_hit(0, [0, 0, 0], [0, 0, -1])          # origin inside: the positive root
    assert h["ok"] and h["t"] == 1.0
    h = kat_scene.prim_hit(0, [0, 0, -5], [0, 0, -1])         # sphere behind the ray: t = -4 is sanitised away
    assert not h["ok"] and h["t"] == FLT_MAX
    h = kat_scene.prim_hit(0, [2, 0, 5], [0, 0, -1])          # miss
    assert not h["ok"]
    h = kat_scene.prim_hit(0, [0, 0, 5], [0, 0, -2])          # direction length 2 halves t
    assert h["ok"] and h["t"] == 2.0


def test_triangle_known_answers(kat_scene):
    n = np.array([0, 0, 1], np.float32)
    h = kat_scene.prim_hit(1, [0.25, 0.25, 1], [0, 0, -1])
    assert h["ok"] and h["t"] == 1.0 and np.array_equal(h["n"], n)
    assert kat_scene.prim_hit(1, [0, 0.5, 1], [0, 0, -1])["ok"]          # beta == 0 is inside
    assert kat_scene.prim_hit(1, [0.5, 0, 1], [0, 0, -1])["ok"]          # gamma == 0 is inside
    assert kat_scene.prim_hit(1, [0.5, 0.5, 1], [0, 0, -1])["ok"]        # beta + gamma == 1 is inside
    assert not kat_scene.prim_hit(1, [0.75, 0.5, 1], [0, 0, -1])["ok"]   # beyond the hypotenuse
    assert not kat_scene.prim_hit(1, [-0.01, 0.5, 1], [0, 0, -1])["ok"]
    assert not kat_scene.prim_hit(1, [0.25, 0.25, -1], [0, 0, -1])["ok"]  # t < 0
    assert not kat_scene.prim_hit(1, [0.25, 0.25, 0], [0, 0, -1])["ok"]   # t == 0
    assert not kat_scene.prim_hit(1, [0.25, 0.25, 1], [1, 0, 0])["ok"]    # parallel: determinant 0, no guard, NaN/inf


def test_container_keeps_first_of_equal_t(tmp_path):
    # two coincident triangles: strict '<' keeps the first in file order (src/geometries.h:184-187)
    p = tmp_path / "dup.test"
    p.write_text("vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\ntri 0 1 2\ntri 0 1 2\n")
    s = RefScene(str(p), use_grid=True)
    h = s.intersect([0.25, 0.25, 1], [0, 0, -1])
    assert h["ok"] and h["prim"] == 0 and h["t"] == 1.0


def test_negative_zero_direction_component_misses_the_grid():
    # D.x = -0.0 is not '< 0', the slabs stay unswapped and the interval is empty (src/grid.h:95-111)
    s = RefScene(scene_path("scene1"))
    assert s.intersect([0, 0, 4], [0.0, 0.0, -1.0])["ok"]
    assert not s.intersect([0, 0, 4], [-0.0, 0.0, -1.0])["ok"]


# --- restated matrix algebra against numpy float64 -------------------------------------------------------------------
def test_inverse_and_transform_stack_against_numpy():
    s = RefScene(scene_path("scene4-specular"))
    p = s.prims()
    for xf, inv in zip(p["xf"], p["inv"]):
        m = xf.reshape(4, 4).T.astype(np.float64)       # column-major storage
        got = inv.reshape(4, 4).T.astype(np.float64)
        assert np.allclose(got, np.linalg.inv(m), rtol=1e-5, atol=1e-6)
    # first table-leg transform of scene4: translate(-1.75,-.8,-.25) translate(0,0,-2) scale(.15,.15,2)
    def T(x, y, z):
        m = np.eye(4); m[:3, 3] = [x, y, z]; return m
    want = T(-1.75, -.8, -.25) @ T(0, 0, -2.0) @ np.diag([0.15, 0.15, 2.0, 1.0])
    got = p["xf"][12].reshape(4, 4).T
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)


def test_rotate_against_numpy(tmp_path):
    p = tmp_path / "rot.test"
    p.write_text("vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nrotate 1 2 3 40\ntri 0 1 2\n")
    s = RefScene(str(p))
    got = s.prims()["xf"][0].reshape(4, 4).T.astype(np.float64)
    a = np.array([1, 2, 3], np.float64); a /= np.linalg.norm(a)
    th = np.deg2rad(40.0)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    assert np.allclose(got[:3, :3], R, atol=1e-6)
    assert np.allclose(got[3], [0, 0, 0, 1]) and np.allclose(got[:3, 3], 0)


# --- committed golden vectors (bit-exact) -----------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["scene1", "scene4-specular", "scene5", "scene6", "scene7"])
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))
    s = RefScene(scene_path(name), counting=True)
    rgb, ids, _, cnt = s.render(int(g["width"]), int(g["height"]), 5)
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(rgb.view(np.int32), g["rgb"].view(np.int32))
    assert np.array_equal(quantize(rgb), g["rgb8"])
    assert [cnt[k] for k in sorted(cnt)] == list(g["counters"])
    bounds, start, cell_ids = s.grid()
    assert np.array_equal(bounds, g["grid_bounds"]) and len(cell_ids) == int(g["grid_refs"])


def test_oracle_reproduces_rows_of_the_4k_golden():
    """The whole-frame 4K golden (tests/golden/make_golden_4k.py, ten CPU-minutes) is too slow to regenerate here; a few
    of its rows through the dragon are, and pin the committed file to the oracle."""
    import hashlib
    g = np.load(os.path.join(GOLDEN, "golden_scene7_4k.npz"))
    W, H = int(g["width"]), int(g["height"])
    assert (W, H, int(g["max_depth"])) == (3840, 2160, 5) and g["ids"].shape == (H, W) and g["rgb8"].shape == (H, W, 3)
    rows = np.arange(1000, H, 400)
    rgb, ids, _, _ = RefScene(scene_path("scene7")).render(W, H, 5, 1000, 400)
    assert np.array_equal(ids[rows], g["ids"][rows]) and (g["ids"][rows] >= 0).mean() > 0.2
    assert np.array_equal(quantize(rgb[rows]), g["rgb8"][rows])
    assert np.array_equal(rgb[rows].astype(np.float64).sum(axis=(1, 2)), g["rgb_row_sums"][rows])
    assert len(str(g["rgb_sha256"])) == len(hashlib.sha256(b"").hexdigest())
    cnt = dict(zip([str(k) for k in g["counter_names"]], [int(v) for v in g["counters"]]))
    hit = int((g["ids"] >= 0).sum())
    assert cnt["primary"] == W * H and cnt["shadow"] >= hit and cnt["reflect"] >= hit      # one light; a mirror ray per hit


def test_ray_counts_follow_the_hit_counts():
    # every hit at depth d spawns n_lights shadow rays, and one mirror ray while depth > 0 (src/main.cpp:85-97)
    s = RefScene(scene_path("scene6"), counting=True)
    _, ids, _, cnt = s.render(80, 60, 5)
    assert cnt["primary"] == 80 * 60
    assert cnt["rays"] == cnt["primary"] + cnt["shadow"] + cnt["reflect"]
    s0 = RefScene(scene_path("scene6"), counting=True)
    _, _, _, c0 = s0.render(80, 60, 0)
    assert c0["reflect"] == 0 and c0["shadow"] == 2 * int((ids >= 0).sum())
