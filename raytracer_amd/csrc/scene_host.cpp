// Host scene layer: the CSE167x-style `.test` scene format flattened into the arrays of
// rtx_scene_view.  Replaces SceneReader::readScene (src/readScene.h:45-182 of the reference tree):
// same directives, same defaults, same right-multiplying transform stack, unknown directives ignored.
// Instead of one heap object per primitive (each with its own Material and mat4 copy, :108-122) the
// parser emits flat arrays with de-duplicated transforms and materials; a primitive's index is its
// position among the `tri` / `sphere` lines, which is the order the reference stores them in.
#include "rtx.h"
#include "rt_xform.h"
#include "rtx_internal.h"

#include <algorithm>
#include <cerrno>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

using rt::Mat4;

struct rtx_scene {
    std::vector<int32_t> prim_kind, prim_xform, prim_material;
    std::vector<float> prim_geom;     // 9 per primitive
    std::vector<float> xforms;        // 16 per transform
    std::vector<float> materials;     // 13 per material
    std::vector<float> lights;        // 10 per light
    rtx_camera cam{};
    int32_t width = 0, height = 0;
    std::string output;
};

namespace {

struct Cursor {
    const char* p;
    const char* end;
};

inline void skip_blank(Cursor& c) {
    while (c.p < c.end && (*c.p == ' ' || *c.p == '\t' || *c.p == '\r' || *c.p == '\v' || *c.p == '\f')) c.p++;
}
// next whitespace-delimited token on the current line; empty when the line is exhausted
inline bool next_token(Cursor& c, const char*& tok, size_t& len) {
    skip_blank(c);
    tok = c.p;
    while (c.p < c.end && !(*c.p == ' ' || *c.p == '\t' || *c.p == '\r' || *c.p == '\v' || *c.p == '\f')) c.p++;
    len = (size_t)(c.p - tok);
    return len > 0;
}
// Formatted-extraction semantics of `linestream >> a >> b >> c` (src/readScene.h:70-170) for one line:
//   * a field that parses is stored;
//   * a field that is present but malformed stores 0 and fails the stream (C++11 num_get);
//   * a field that is missing stores nothing and fails the stream (the sentry fails at end of line);
//   * once the stream has failed, every later extraction stores nothing.
// A number followed by junk ("1.5abc") stores the number and leaves the junk in the stream, where the next
// extraction finds it malformed.  The state directives (ambient, diffuse, ..., camera, size) extract straight into
// the state, so a short or malformed line keeps the earlier values of the fields it does not reach; directives that
// extract into locals (vertex, tri, sphere, translate, ...) read indeterminate values there, for which this parser
// takes 0.
struct Fields {
    Cursor c;
    bool failed = false;
    // The prefix of a token that the stream accumulates for a floating-point field before converting it:
    // [sign] digits [. digits] [e|E [sign] digits]  (any part may be empty; the conversion then rejects it).
    static size_t number_prefix(const char* t, size_t n) {
        size_t m = 0;
        if (m < n && (t[m] == '+' || t[m] == '-')) m++;
        while (m < n && t[m] >= '0' && t[m] <= '9') m++;
        if (m < n && t[m] == '.') { m++; while (m < n && t[m] >= '0' && t[m] <= '9') m++; }
        if (m < n && (t[m] == 'e' || t[m] == 'E')) {
            m++;
            if (m < n && (t[m] == '+' || t[m] == '-')) m++;
            while (m < n && t[m] >= '0' && t[m] <= '9') m++;
        }
        return m;
    }
    bool begin(const char*& t, size_t& n) {
        if (failed) return false;
        if (!next_token(c, t, n)) { failed = true; return false; }   // missing: nothing is stored
        return true;
    }
    // Decimal -> float, correctly rounded as the stream extraction is.  std::from_chars covers the plain decimal
    // spellings scene files use; anything it does not take whole (a leading '+', trailing characters) goes through
    // strtof on the prefix the stream would have accumulated.
    bool f(float& dst) {
        const char* t;
        size_t n;
        if (!begin(t, n)) return false;
        float v = 0.0f;
        const std::from_chars_result r = std::from_chars(t, t + n, v);
        if (r.ec == std::errc() && r.ptr == t + n && number_prefix(t, n) == n) { dst = v; return true; }
        const size_t m = std::min(number_prefix(t, n), (size_t)63);
        if (m > 0) {
            char buf[64];
            memcpy(buf, t, m);
            buf[m] = 0;
            char* e = nullptr;
            v = strtof(buf, &e);
            if (e == buf + m) {                                         // the accumulated characters are a number
                dst = v;
                c.p = t + m;                                            // the rest of the token stays in the stream
                return true;
            }
        }
        dst = 0.0f;                                                     // malformed
        failed = true;
        return false;
    }
    bool i(long& dst) {
        const char* t;
        size_t n;
        if (!begin(t, n)) return false;
        long v = 0;
        const std::from_chars_result r = std::from_chars(t, t + n, v, 10);
        if (r.ec == std::errc() && r.ptr == t + n) { dst = v; return true; }
        char buf[64];
        size_t m = 0;
        if (m < n && (t[m] == '+' || t[m] == '-')) buf[m] = t[m], m++;
        while (m < n && m + 1 < sizeof buf && t[m] >= '0' && t[m] <= '9') { buf[m] = t[m]; m++; }
        buf[m] = 0;
        char* e = nullptr;
        v = strtol(buf, &e, 10);
        if (m > 0 && e == buf + m) {
            dst = v;
            c.p = t + m;
            return true;
        }
        dst = 0;
        failed = true;
        return false;
    }
    // extraction into a local that starts at 0
    float f() { float v = 0.0f; f(v); return v; }
    long i() { long v = 0; i(v); return v; }
};

struct MatKey {
    float v[16];
    bool operator==(const MatKey& o) const { return memcmp(v, o.v, sizeof v) == 0; }
};
struct MatKeyHash {
    size_t operator()(const MatKey& k) const {
        uint64_t h = 1469598103934665603ull;
        const unsigned char* b = (const unsigned char*)k.v;
        for (size_t i = 0; i < sizeof k.v; i++) { h ^= b[i]; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

bool cmd_is(const char* t, size_t n, const char* name) { return strlen(name) == n && memcmp(t, name, n) == 0; }

int parse_text(const char* text, size_t n_bytes, rtx_scene* sc) {
    // state captured at primitive / light creation (src/readScene.h:46-51)
    float diffuse[3] = {1, 1, 1}, ambient[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
    float atten[3] = {1, 0, 0};
    float shininess = 40;
    std::vector<Mat4> stack;
    stack.push_back(rt::mat_identity());
    std::vector<float> verts;

    std::unordered_map<MatKey, int32_t, MatKeyHash> xform_ids, material_ids;
    auto intern = [](std::unordered_map<MatKey, int32_t, MatKeyHash>& map, std::vector<float>& store,
                     const float* v, int n) -> int32_t {
        MatKey k;
        memset(&k, 0, sizeof k);
        memcpy(k.v, v, n * sizeof(float));
        auto it = map.find(k);
        if (it != map.end()) return it->second;
        int32_t id = (int32_t)(store.size() / n);
        store.insert(store.end(), v, v + n);
        map.emplace(k, id);
        return id;
    };
    // the current material / transform is looked up again only after a directive changed it: a mesh of
    // 100 000 triangles under one material and one transform costs two lookups, not 200 000
    int32_t material_id = -1, xform_id = -1;
    auto cur_material = [&]() {
        if (material_id < 0) {
            float m[13] = {ambient[0], ambient[1], ambient[2], diffuse[0], diffuse[1], diffuse[2],
                           specular[0], specular[1], specular[2], emission[0], emission[1], emission[2], shininess};
            material_id = intern(material_ids, sc->materials, m, 13);
        }
        return material_id;
    };
    auto cur_xform = [&]() {
        if (xform_id < 0) xform_id = intern(xform_ids, sc->xforms, stack.back().m, 16);
        return xform_id;
    };

    const char* p = text;
    const char* end = text + n_bytes;
    long line_no = 0;
    while (p < end) {
        const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        line_no++;
        Fields fl{{p, eol}};
        p = eol + 1;
        const char* t;
        size_t n;
        if (!next_token(fl.c, t, n)) continue;
        if (t[0] == '#') continue;

        if (cmd_is(t, n, "vertex")) {
            float x = fl.f(), y = fl.f(), z = fl.f();
            verts.push_back(x); verts.push_back(y); verts.push_back(z);
        } else if (cmd_is(t, n, "tri")) {
            long a = fl.i(), b = fl.i(), c = fl.i();
            long nv = (long)(verts.size() / 3);
            if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) {
                rtx_set_error("line %ld: tri index out of range (%ld %ld %ld, %ld vertices)", line_no, a, b, c, nv);
                return RTX_E_INVALID;
            }
            sc->prim_kind.push_back(RTX_PRIM_TRIANGLE);
            const size_t at = sc->prim_geom.size();
            sc->prim_geom.resize(at + 9);
            memcpy(&sc->prim_geom[at], &verts[3 * a], 3 * sizeof(float));
            memcpy(&sc->prim_geom[at + 3], &verts[3 * b], 3 * sizeof(float));
            memcpy(&sc->prim_geom[at + 6], &verts[3 * c], 3 * sizeof(float));
            sc->prim_material.push_back(cur_material());
            sc->prim_xform.push_back(cur_xform());
        } else if (cmd_is(t, n, "sphere")) {
            float x = fl.f(), y = fl.f(), z = fl.f(), r = fl.f();
            sc->prim_kind.push_back(RTX_PRIM_SPHERE);
            const float g[9] = {x, y, z, r, 0, 0, 0, 0, 0};
            sc->prim_geom.insert(sc->prim_geom.end(), g, g + 9);
            sc->prim_material.push_back(cur_material());
            sc->prim_xform.push_back(cur_xform());
        } else if (cmd_is(t, n, "size")) {
            long w = sc->width, h = sc->height;
            fl.i(w); fl.i(h);
            sc->width = (int32_t)w; sc->height = (int32_t)h;
        } else if (cmd_is(t, n, "camera")) {
            for (int k = 0; k < 3; k++) fl.f(sc->cam.eye[k]);
            for (int k = 0; k < 3; k++) fl.f(sc->cam.center[k]);
            for (int k = 0; k < 3; k++) fl.f(sc->cam.up[k]);
            fl.f(sc->cam.fov_deg);
        } else if (cmd_is(t, n, "point") || cmd_is(t, n, "directional")) {
            const bool point = cmd_is(t, n, "point");
            float l[10];
            l[0] = point ? (float)RTX_LIGHT_POINT : (float)RTX_LIGHT_DIRECTIONAL;
            for (int k = 1; k <= 6; k++) l[k] = fl.f();
            // attenuation is captured for point lights only (src/readScene.h:83-86); a directional light
            // never reads it (src/main.cpp:67-75), so it gets the neutral 1 0 0
            l[7] = point ? atten[0] : 1.0f;
            l[8] = point ? atten[1] : 0.0f;
            l[9] = point ? atten[2] : 0.0f;
            sc->lights.insert(sc->lights.end(), l, l + 10);
        } else if (cmd_is(t, n, "ambient")) { for (int k = 0; k < 3; k++) fl.f(ambient[k]); material_id = -1; }
        else if (cmd_is(t, n, "diffuse")) { for (int k = 0; k < 3; k++) fl.f(diffuse[k]); material_id = -1; }
        else if (cmd_is(t, n, "specular")) { for (int k = 0; k < 3; k++) fl.f(specular[k]); material_id = -1; }
        else if (cmd_is(t, n, "emission")) { for (int k = 0; k < 3; k++) fl.f(emission[k]); material_id = -1; }
        else if (cmd_is(t, n, "shininess")) { fl.f(shininess); material_id = -1; }
        else if (cmd_is(t, n, "attenuation")) { for (int k = 0; k < 3; k++) fl.f(atten[k]); }
        else if (cmd_is(t, n, "pushTransform")) { stack.push_back(stack.back()); }
        else if (cmd_is(t, n, "popTransform")) { if (stack.size() > 1) stack.pop_back(); xform_id = -1; }   // identity is unpoppable
        else if (cmd_is(t, n, "translate")) {
            float x = fl.f(), y = fl.f(), z = fl.f();
            stack.back() = rt::mat_translate(stack.back(), x, y, z);
            xform_id = -1;
        } else if (cmd_is(t, n, "rotate")) {
            float x = fl.f(), y = fl.f(), z = fl.f(), deg = fl.f();
            const float pi = 3.14159265358979323846264338327950288f;
            stack.back() = rt::mat_mul(stack.back(), rt::mat_rotate(rt::mat_identity(), deg * pi / 180.f, x, y, z));
            xform_id = -1;
        } else if (cmd_is(t, n, "scale")) {
            float x = fl.f(), y = fl.f(), z = fl.f();
            stack.back() = rt::mat_mul(stack.back(), rt::mat_scale(rt::mat_identity(), x, y, z));
            xform_id = -1;
        } else if (cmd_is(t, n, "output")) {
            const char* o;
            size_t on;
            if (next_token(fl.c, o, on)) sc->output.assign(o, on);
        }
        // everything else (maxdepth, maxverts, comments without a space, ...) is ignored on purpose
    }
    return RTX_OK;
}

}  // namespace

extern "C" {

int rtx_scene_parse(rtx_scene** out, const char* text, int64_t n_bytes) {
    if (!out || (!text && n_bytes > 0) || n_bytes < 0) { rtx_set_error("rtx_scene_parse: bad argument"); return RTX_E_INVALID; }
    rtx_scene* sc = new (std::nothrow) rtx_scene;
    if (!sc) return RTX_E_NOMEM;
    int rc;
    try {
        rc = parse_text(text, (size_t)n_bytes, sc);
    } catch (const std::bad_alloc&) {
        rc = RTX_E_NOMEM;
    }
    if (rc != RTX_OK) { delete sc; *out = nullptr; return rc; }
    *out = sc;
    return RTX_OK;
}

int rtx_scene_load(rtx_scene** out, const char* path) {
    if (!out || !path) { rtx_set_error("rtx_scene_load: bad argument"); return RTX_E_INVALID; }
    FILE* f = fopen(path, "rb");
    if (!f) { rtx_set_error("cannot open scene file '%s': %s", path, strerror(errno)); return RTX_E_IO; }
    std::string buf;
    char chunk[1 << 16];
    size_t got;
    try {
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, got);
    } catch (const std::bad_alloc&) {
        fclose(f);
        return RTX_E_NOMEM;
    }
    fclose(f);
    return rtx_scene_parse(out, buf.data(), (int64_t)buf.size());
}

void rtx_scene_free(rtx_scene* s) { delete s; }

int rtx_scene_get_view(const rtx_scene* s, rtx_scene_view* v) {
    if (!s || !v) { rtx_set_error("rtx_scene_get_view: bad argument"); return RTX_E_INVALID; }
    v->n_prims = (int32_t)s->prim_kind.size();
    v->n_xforms = (int32_t)(s->xforms.size() / 16);
    v->n_materials = (int32_t)(s->materials.size() / 13);
    v->n_lights = (int32_t)(s->lights.size() / 10);
    v->prim_kind = s->prim_kind.data();
    v->prim_geom = s->prim_geom.data();
    v->prim_xform = s->prim_xform.data();
    v->prim_material = s->prim_material.data();
    v->xforms = s->xforms.data();
    v->materials = s->materials.data();
    v->lights = s->lights.data();
    return RTX_OK;
}

int rtx_scene_get_camera(const rtx_scene* s, rtx_camera* out) {
    if (!s || !out) { rtx_set_error("rtx_scene_get_camera: bad argument"); return RTX_E_INVALID; }
    *out = s->cam;
    return RTX_OK;
}

int rtx_scene_get_size(const rtx_scene* s, int32_t* w, int32_t* h) {
    if (!s || !w || !h) { rtx_set_error("rtx_scene_get_size: bad argument"); return RTX_E_INVALID; }
    *w = s->width;
    *h = s->height;
    return RTX_OK;
}

const char* rtx_scene_get_output(const rtx_scene* s) { return s ? s->output.c_str() : ""; }

}  // extern "C"
