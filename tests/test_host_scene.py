"""Host scene layer of librtx (parser, transform stack, image output, row split) against the CPU oracle.  No GPU."""
import os

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as rt
from conftest import scene_path
from oracle import ref_oracle
from oracle.ref_oracle import RefScene
from raytracer_amd.multi_gpu import rank_rows

ALL_SCENES = ["scene1", "scene2", "scene3", "scene4-ambient", "scene4-diffuse", "scene4-emission", "scene4-specular",
              "scene5", "scene6", "scene7", "myScene2"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


@pytest.mark.parametrize("name", ALL_SCENES)
def test_parser_matches_oracle_bit_for_bit(name):
    ref = RefScene(scene_path(name))
    p = ref.prims()
    sr = rt.SceneReader().readScene(scene_path(name))
    assert sr.n_prims == ref.n_prims and (sr.width, sr.height) == (ref.width, ref.height)
    assert sr.outputFilename == ref.output
    assert np.array_equal(sr.prim_kind, p["kind"])
    assert np.array_equal(bits(sr.prim_geom), bits(p["geom"]))
    assert np.array_equal(bits(sr.xforms[sr.prim_xform]), bits(p["xf"]))          # transform stack, bit for bit
    assert np.array_equal(bits(sr.materials[sr.prim_material]), bits(p["mat"]))
    assert np.array_equal(bits(sr.lights), bits(ref.lights()))
    cam = ref.camera()
    for k in ("eye", "center", "up"):
        assert np.array_equal(bits(sr.camera[k]), bits(cam[k]))
    assert sr.camera["fov"] == cam["fov"]


def test_transforms_and_materials_are_deduplicated():
    sr = rt.SceneReader().readScene(scene_path("scene7"))
    assert sr.n_prims == 100000 and len(sr.xforms) == 1 and len(sr.materials) == 1
    sr = rt.SceneReader().readScene(scene_path("scene5"))
    assert sr.n_prims == 1000 and len(sr.xforms) == 1000 and len(sr.materials) == 1


def test_directives_defaults_and_ignored_lines():
    sr = rt.SceneReader().parse("# comment\n#camera 1 2 3\nmaxdepth 9\nmaxverts 3\nbogus 1 2 3\n"
                                "vertex 0 0 0\nvertex +1 -1 .5\nvertex 1e0 2E-1 3\ntri 0 1 2\n"
                                "popTransform\npopTransform\n"       # the identity at the bottom cannot be popped
                                "sphere 1 2 3 4\n")
    assert sr.n_prims == 2
    assert list(sr.prim_geom[0]) == [0, 0, 0, 1, -1, .5, 1, np.float32(0.2), 3]
    # defaults: ambient 0, diffuse 1, specular 0, emission 0, shininess 40 (src/readScene.h:46-51)
    assert list(sr.materials[0]) == [0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 40]
    assert np.array_equal(sr.xforms[0].reshape(4, 4), np.eye(4, dtype=np.float32))
    assert (sr.width, sr.height, sr.outputFilename) == (0, 0, "")


def test_material_and_attenuation_are_captured_at_creation():
    sr = rt.SceneReader().parse("vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\n"
                                "point 1 2 3 .1 .2 .3\nattenuation 1 .5 .25\npoint 4 5 6 1 1 1\ndirectional 0 0 1 1 1 1\n"
                                "diffuse .5 .5 .5\ntri 0 1 2\nshininess 7\ntri 0 1 2\ndiffuse .5 .5 .5\ntri 0 1 2\n")
    assert [list(l[7:]) for l in sr.lights] == [[1, 0, 0], [1, .5, .25], [1, 0, 0]]
    assert [int(l[0]) for l in sr.lights] == [0, 0, 1]
    assert list(sr.prim_material) == [0, 1, 1] and sr.materials[1][12] == 7


def test_missing_fields_read_as_zero_and_poison_the_line():
    sr = rt.SceneReader().parse("vertex 1 2\nvertex 1 x 3\nvertex 0 1 0\ntri 0 1 2\n")
    assert list(sr.prim_geom[0]) == [1, 2, 0, 1, 0, 0, 0, 1, 0]


SHORT_LINE_SCENE = """size 640 480
size 320
camera 1 2 3 4 5 6 0 1 0 45
camera 9 8
ambient .1 .2 .3
ambient .7
diffuse .5 .6 .7
diffuse x .9 .9
specular .2 .3 .4
specular .8 oops .9
emission .01 .02 .03
emission
shininess 12
shininess
attenuation 1 .5 .25
attenuation 2
point 1 2 3 1 1 1
attenuation 3 .1abc .7
point 4 5 6 1 1 1
vertex 0 0 0
vertex 1 0 0
vertex 0 1 0
tri 0 1 2
shininess 1.5e
diffuse 1e1 +2 3.
tri 0 1 2
ambient .25.5 7
tri 0 1 2
"""


def test_short_and_malformed_state_lines_keep_what_the_stream_keeps(tmp_path):
    """`linestream >> a >> b >> c` (src/readScene.h:70-170): a missing field and every field after a failure leave their
    targets as they were, the first malformed field reads 0, a number followed by junk leaves the junk for the next
    field.  No scene the reference ships has such lines; the oracle parses with the same stream calls."""
    path = tmp_path / "short.test"
    path.write_text(SHORT_LINE_SCENE)
    ref = RefScene(str(path))
    sr = rt.SceneReader().readScene(str(path))
    cam = ref.camera()
    assert (sr.width, sr.height) == (ref.width, ref.height) == (320, 480)
    for k in ("eye", "center", "up"):
        assert np.array_equal(bits(sr.camera[k]), bits(cam[k]))
    assert list(sr.camera["eye"]) == [9, 8, 3] and sr.camera["fov"] == cam["fov"] == 45
    assert np.array_equal(bits(sr.materials[sr.prim_material]), bits(ref.prims()["mat"]))
    assert np.array_equal(bits(sr.lights), bits(ref.lights()))
    m = sr.materials[sr.prim_material]
    f = np.float32
    assert list(m[0]) == [f(.7), f(.2), f(.3), 0, f(.6), f(.7), f(.8), 0, f(.4), f(.01), f(.02), f(.03), 12]
    assert m[1][12] == 0 and list(m[1][3:6]) == [10, 2, 3]           # "1.5e" is malformed: 0
    assert list(m[2][0:3]) == [f(.25), f(.5), 7]                      # ".25.5" is two numbers: .25 and .5
    assert [list(l[7:]) for l in sr.lights] == [[2, f(.5), f(.25)], [3, f(.1), 0]]      # "abc" is what the third field finds: 0


def test_errors_are_codes_not_crashes(tmp_path):
    with pytest.raises(rt.RtxError) as e:
        rt.SceneReader().readScene(str(tmp_path / "nope.test"))
    assert e.value.code == -2
    with pytest.raises(rt.RtxError) as e:
        rt.SceneReader().parse("vertex 0 0 0\ntri 0 1 2\n")
    assert e.value.code == -1 and "out of range" in str(e.value)
    with pytest.raises(rt.RtxError):
        rt.SceneReader().parse("vertex 0 0 0\ntri 0 0 -1\n")
    assert rt.SceneReader().parse("").n_prims == 0


def test_quantize_truncates_like_the_reference():
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-0.5, 1.5, 100000).astype(np.float32),
                        np.array([0, 1, 0.999999, 1 / 255, 254.999 / 255, -0.0, 2, -3, np.nan], np.float32),
                        (np.arange(256) / 255.0).astype(np.float32)])
    got = rt.quantize(x)
    ok = ~np.isnan(x)
    assert np.array_equal(got[ok], ref_oracle.quantize(x)[ok])
    assert got[np.isnan(x)][0] == 0
    want = np.floor(np.clip(x[ok].astype(np.float32), 0, 1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(got[ok], want)


def test_png_and_ppm_writers(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.uniform(-0.2, 1.2, (37, 53, 3)).astype(np.float32)
    rt.save_image(str(tmp_path / "a.png"), img)
    rt.save_image(str(tmp_path / "a.ppm"), img)
    want = rt.quantize(img)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png").convert("RGB")), want)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.ppm").convert("RGB")), want)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n53 37\n255\n") and len(raw) == len(b"P6\n53 37\n255\n") + 37 * 53 * 3
    with pytest.raises(rt.RtxError):
        rt.save_image(str(tmp_path / "no_such_dir" / "a.png"), img)
    im = rt.Image3f(4, 3)
    im.setAt(1, 2, (0.5, 0.25, 1.0))
    assert list(im.getAt(1, 2)) == [0.5, 0.25, 1.0]
    im.save(str(tmp_path / "b.png"))
    assert tuple(np.asarray(Image.open(tmp_path / "b.png"))[2, 1]) == (127, 63, 255)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (5000, 3), (3, 5000)])
def test_png_of_a_large_frame_is_compressed_in_bands_and_decodes_to_the_same_pixels(tmp_path, w, h):
    """Frames above a megabyte are deflated as several bands of rows side by side, laid end to end in one stream."""
    rng = np.random.default_rng(w + h)
    img = np.zeros((h, w, 3), np.uint8)
    img[h // 4: h // 2 + 1] = rng.integers(0, 256, (h // 2 + 1 - h // 4, w, 3), dtype=np.uint8)      # noise beside flat rows
    img[:, w // 3] = 200
    rt.save_image_u8(str(tmp_path / "big.png"), img)
    back = np.asarray(Image.open(tmp_path / "big.png").convert("RGB"))
    assert back.shape == img.shape and np.array_equal(back, img)


@pytest.mark.parametrize("height,tile_rows,world", [(2160, 8, 8), (2160, 8, 1), (480, 8, 3), (23, 5, 4), (7, 8, 2), (1, 1, 1),
                                                    (100, 16, 7)])
def test_row_split_is_a_partition(height, tile_rows, world):
    seen = []
    for rank in range(world):
        rows = rt.local_rows(height, tile_rows, rank, world)
        assert np.array_equal(rows, rank_rows(height, tile_rows, rank, world))
        assert np.all(np.diff(rows) > 0)
        seen.extend(rows.tolist())
    assert sorted(seen) == list(range(height))


def test_row_split_rejects_bad_rank():
    with pytest.raises(rt.RtxError):
        rt.local_rows(10, 8, 2, 2)
    with pytest.raises(rt.RtxError):
        rt.local_rows(0, 8, 0, 1)
