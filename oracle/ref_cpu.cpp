// TEST INFRASTRUCTURE ONLY -- CPU oracle for the per-pixel trace loop.
//
// A CPU restatement of the reference renderer's algorithm, kept in the
// reference's own shape (heap-allocated primitives behind virtual calls, a
// matrix inverse per primitive test, fixed 15^3 grid, row-parallel OpenMP) so
// that it serves both as the parity checker and as the timed CPU baseline.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; the product (raytracer_amd/) never does.
//
// PARITY UNPINNED at the bit level: the reference cannot be built in this image
// (glm and OpenCV are absent) and it ships no tests or golden vectors; the only
// fixtures it holds are eight lossy course JPEGs (res/scene1-camera1..4.jpg,
// res/scene2-camera1..3.jpg, res/scene3.jpg), against which this oracle is
// pinned coarsely, each under the camera its scene file lists for it
// (silhouette / mean-error checks in tests/test_oracle_fixtures.py).
//
// Each function cites the reference lines it follows (paths relative to
// /root/reference).
#include "ref_math.h"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <stack>
#include <string>
#include <vector>

#include <omp.h>

using namespace refm;

namespace {

// src/geometries.h:21-25
const float F_MAX = std::numeric_limits<float>::max();
const float F_MIN = std::numeric_limits<float>::min();   // smallest POSITIVE normal: seeds every box maximum
inline float positive_or_max(float t) { return t > 0 ? t : F_MAX; }

#ifdef REF_COUNT
struct Counters {
    uint64_t rays, tri_tests, sph_tests, cell_steps, light_evals, hits, primary, shadow, reflect;
};
static thread_local Counters tl_cnt;
#define CNT(field, n) (tl_cnt.field += (n))
#else
#define CNT(field, n) ((void)0)
#endif

// src/geometries.h:27-36
V3 xf_normal(const M4& m, V3 n) { V4 r = transpose(inverse(m)) * V4(n, 0); return normalize(V3(r.x, r.y, r.z)); }
V3 xf_point(const M4& m, V3 p) { V4 r = m * V4(p, 1); return V3(r.x, r.y, r.z); }
V3 xf_dir(const M4& m, V3 d) { V4 r = m * V4(d, 0); return V3(r.x, r.y, r.z); }

// src/geometries.h:38-56
struct Mat {
    V3 ambient, diffuse, specular, emission;
    float shininess = 0;
};

// src/geometries.h:59-78 (object-space hit) -- the sanitising constructor
struct OHit {
    bool ok = false;
    float t = F_MAX;
    V3 n;
    OHit() {}
    OHit(bool v, float tt, V3 nn) : ok(tt > 0 ? v : false), t(positive_or_max(tt)), n(nn) {}
};

// src/geometries.h:81-104 (world-space hit); `prim` stands in for the Material* identity
struct WHit {
    bool ok = false;
    float t = F_MAX;
    V3 p, n;
    int prim = -1;
};

struct Box { V3 lo, hi; };

// src/geometries.h:112-156
struct Prim {
    Mat mat;
    M4 xf;
    int id = -1;
    virtual ~Prim() {}
    virtual OHit hit(V3 O, V3 D) = 0;
    virtual Box extent() = 0;

    // :135-155 -- maximum seeded with FLOAT_MIN (a tiny positive), reproduced on purpose
    Box world_box(V3 lo, V3 hi) {
        V3 pick[2] = {lo, hi};
        V3 mn = V3(1, 1, 1) * F_MAX, mx = V3(1, 1, 1) * F_MIN;
        for (int i = 0; i < 8; i++) {
            int a = (i / 1) % 2, b = (i / 2) % 2, c = (i / 4) % 2;   // :127-130
            V3 p = xf_point(xf, V3(pick[a].x, pick[b].y, pick[c].z));
            mn = gmin(mn, p);
            mx = gmax(mx, p);
        }
        return {mn, mx};
    }
};

// src/geometries.h:209-260
struct Ball : Prim {
    V3 S;
    float r;
    OHit hit(V3 O, V3 D) override {
        CNT(sph_tests, 1);
        V3 Q = O - S;
        float p = 2 * dot(Q, D) / dot(D, D);
        float q = (dot(Q, Q) - r * r) / dot(D, D);
        float disc = p * p / 4 - q;
        if (disc == 0) {
            float t = -p / 2;
            return OHit(true, t, (O + t * D) - S);
        } else if (disc > 0) {
            float root = std::sqrt(disc);
            float x1 = -p / 2 + root, x2 = -p / 2 - root;
            float t = gmin(x1, x2);
            if (x1 <= 0) t = x2;
            if (x2 <= 0) t = x1;
            return OHit(true, t, (O + t * D) - S);
        }
        return OHit();
    }
    Box extent() override {
        V3 d(r, r, r);
        return world_box(S - d, S + d);
    }
};

// src/geometries.h:262-319
struct Tri : Prim {
    V3 A, B, C;
    static float term(V3 a, V3 b, V3 c, int k) {   // :267-272 rule of Sarrus, products left to right
        int i1 = k % 3, i2 = (k + 1) % 3, i3 = (k + 2) % 3;
        return a[i1] * b[i2] * c[i3] - c[i1] * b[i2] * a[i3];
    }
    static float det(V3 a, V3 b, V3 c) {           // :275-280
        float t1 = term(a, b, c, 0), t2 = term(a, b, c, 1), t3 = term(a, b, c, 2);
        return t1 + t2 + t3;
    }
    static V3 solve(V3 a, V3 b, V3 c, V3 d) {      // :288-294 Cramer, three true divides, no det==0 guard
        float dA = det(a, b, c);
        return V3(det(d, b, c) / dA, det(a, d, c) / dA, det(a, b, d) / dA);
    }
    OHit hit(V3 O, V3 D) override {                // :305-311
        CNT(tri_tests, 1);
        V3 s = solve(A - B, A - C, D, A - O);
        if (s.x >= 0 && s.y >= 0 && s.x + s.y <= 1 && s.z > 0)
            return OHit(true, s.z, normalize(cross(B - A, C - A)));
        return OHit();
    }
    Box extent() override { return world_box(gmin(gmin(A, B), C), gmax(gmax(A, B), C)); }   // :313-318
};

struct Accel {
    virtual ~Accel() {}
    virtual WHit hit(V3 O, V3 D, float t_limit = FLT_MAX) = 0;
};

// src/geometries.h:158-205 -- brute-force list; the inverse is recomputed per primitive per ray
struct Bag : Accel {
    std::vector<Prim*> items;
    WHit hit(V3 O, V3 D, float t_limit = FLT_MAX) override {
        OHit best;
        Prim* who = nullptr;
        for (Prim* g : items) {
            V3 Oo = xf_point(inverse(g->xf), O);
            V3 Do = xf_dir(inverse(g->xf), D);
            OHit h = g->hit(Oo, Do);
            if (h.ok && h.t < best.t && h.t < t_limit) { best = h; who = g; }
        }
        WHit w;
        if (best.ok) {
            CNT(hits, 1);
            w.ok = true;
            w.t = best.t;
            w.p = O + best.t * D;
            w.n = xf_normal(who->xf, best.n);
            w.prim = who->id;
        }
        return w;
    }
};

// src/Image3f.h:15-21
inline float clamp01(float v, float lo = 0.0f, float hi = 1.0f) { return gmax(gmin(v, hi), lo); }
inline V3 clamp_rgb(V3 c) { return V3(clamp01(c[0]), clamp01(c[1]), clamp01(c[2])); }

inline float max3(float a, float b, float c) { float m = a; if (m < b) m = b; if (m < c) m = c; return m; }   // std::max({..})
inline float min3(float a, float b, float c) { float m = a; if (b < m) m = b; if (c < m) m = c; return m; }
inline float min4(float a, float b, float c, float d) { float m = min3(a, b, c); if (d < m) m = d; return m; }

// src/grid.h:11-266
struct Lattice : Accel {
    V3 lo, hi, size, res;
    std::unique_ptr<Bag[]> cells;

    explicit Lattice(std::vector<Prim*>& prims) {     // :36-49
        res = V3(1, 1, 1) * 15.0f;
        // :21-34 scene bounds, same FLOAT_MIN seed, then +-0.001
        V3 mn = V3(1, 1, 1) * F_MAX, mx = V3(1, 1, 1) * F_MIN;
        for (Prim* g : prims) {
            Box b = g->extent();
            mn = gmin(gmin(mn, b.lo), b.hi);
            mx = gmax(gmax(mx, b.hi), b.lo);
        }
        const float eps = 0.001;
        V3 e = V3(1, 1, 1) * eps;
        lo = mn - e;
        hi = mx + e;
        size = hi - lo;
        cells.reset(new Bag[int(res.x) * int(res.y) * int(res.z)]);
        for (Prim* g : prims) {                       // :76-88, file order preserved per cell
            Box b = g->extent();
            int x0, y0, z0, x1, y1, z1;
            cell_of(b.lo, x0, y0, z0);
            cell_of(b.hi, x1, y1, z1);
            for (int z = z0; z <= z1; z++)
                for (int y = y0; y <= y1; y++)
                    for (int x = x0; x <= x1; x++) cells[offset(x, y, z)].items.push_back(g);
        }
    }
    void cell_of(V3 p, int& ix, int& iy, int& iz) {   // :53-59
        ix = clamp01(std::floor((p.x - lo.x) * res.x / size.x), 0, res.x - 1);
        iy = clamp01(std::floor((p.y - lo.y) * res.y / size.y), 0, res.y - 1);
        iz = clamp01(std::floor((p.z - lo.z) * res.z / size.z), 0, res.z - 1);
    }
    int offset(int x, int y, int z) { return x + res.x * y + res.y * res.x * z; }   // :61-63 (float arithmetic)

    bool inside(V3 p) {                               // :140-147
        return p.x > lo.x && p.x < hi.x && p.y > lo.y && p.y < hi.y && p.z > lo.z && p.z < hi.z;
    }

    WHit hit(V3 O, V3 D, float t_limit = FLT_MAX) override {   // :149-261
        CNT(rays, 1);
        // :91-138 slab test; relies on IEEE inf for zero direction components
        const float eps = 0.0001;
        float txs = (lo.x - O.x) / D.x, txe = (hi.x - O.x) / D.x;
        float tys = (lo.y - O.y) / D.y, tye = (hi.y - O.y) / D.y;
        float tzs = (lo.z - O.z) / D.z, tze = (hi.z - O.z) / D.z;
        float tx0 = txs, tx1 = txe; if (D.x < 0) { tx0 = txe; tx1 = txs; }
        float ty0 = tys, ty1 = tye; if (D.y < 0) { ty0 = tye; ty1 = tys; }
        float tz0 = tzs, tz1 = tze; if (D.z < 0) { tz0 = tze; tz1 = tzs; }
        float t0 = max3(tx0, ty0, tz0), t1 = min3(tx1, ty1, tz1);
        V3 dt((tx1 - tx0) / res.x, (ty1 - ty0) / res.y, (tz1 - tz0) / res.z);
        if (!(t0 < t1 && t1 > eps)) return WHit();

        V3 pos = inside(O) ? O : O + t0 * D;          // :156-162
        int ix, iy, iz;
        cell_of(pos, ix, iy, iz);

        int sx, ex, sy, ey, sz, ez;
        float nx, ny, nz;
        if (D.x > 0) { sx = 1; ex = int(res.x); nx = tx0 + (ix + 1) * dt.x; }            // :166-185
        else if (D.x < 0) { sx = -1; ex = -1; nx = tx0 + (int(res.x) - ix) * dt.x; }
        else { sx = -1; ex = -1; nx = F_MAX; }
        if (D.y > 0) { sy = 1; ey = int(res.y); ny = ty0 + (iy + 1) * dt.y; }            // :187-206
        else if (D.y < 0) { sy = -1; ey = -1; ny = ty0 + (int(res.y) - iy) * dt.y; }
        else { sy = -1; ey = -1; ny = F_MAX; }
        if (D.z > 0) { sz = 1; ez = int(res.z); nz = tz0 + (iz + 1) * dt.z; }            // :208-227
        else if (D.z < 0) { sz = -1; ez = -1; nz = tz0 + (int(res.z) - iz) * dt.z; }
        else { sz = -1; ez = -1; nz = F_MAX; }

        while (ix != ex && iy != ey && iz != ez) {    // :229-254
            CNT(cell_steps, 1);
            float t_exit = min4(nx, ny, nz, t_limit);
            WHit h = cells[offset(ix, iy, iz)].hit(O, D, t_exit);
            if (h.ok) return h;
            if (t_exit == nx) { ix += sx; nx += dt.x; }
            else if (t_exit == ny) { iy += sy; ny += dt.y; }
            else if (t_exit == nz) { iz += sz; nz += dt.z; }
            else if (t_exit == t_limit) return WHit();
        }
        return WHit();
    }
};

// src/geometries.h:321-364
struct Cam {
    V3 eye, center, up;
    float fov = 0;
    V3 u, v, w;
    int width = 0, height = 0;
    void axes() {                                     // :332-337
        V3 dir = center - eye;
        w = normalize(-dir);
        u = normalize(cross(up, w));
        v = normalize(cross(w, u));
    }
    V3 ray(int x, int y) {                            // :339-351
        float j = (float)x + 0.5;
        float i = (float)y + 0.5;
        const float aspect = (float)width / (float)height;
        const float tany = std::tan(fov * PI_F / 360.f);
        const float tanx = tany * aspect;
        const float halfW = (float)width / 2.f, halfH = (float)height / 2.f;
        const float a = tanx * ((j - halfW) / halfW);
        const float b = tany * ((halfH - i) / halfH);
        return normalize(a * u + b * v - w);
    }
};
enum LightKind { POINT_L = 0, DIR_L = 1 };
struct Lamp { V3 pos, color; int kind; V3 atten; };

// src/readScene.h:25-183
struct Scene {
    Cam cam;
    std::vector<Lamp> lamps;
    std::vector<V3> verts;
    std::vector<Prim*> prims;
    std::vector<M4> prim_xf;                          // kept for the dump API only
    std::unique_ptr<Accel> accel;
    Lattice* lattice = nullptr;
    std::string output;
    const float bias = 0.001f;
    bool opened = false;
    ~Scene() { for (Prim* p : prims) delete p; }

    void read(const std::string& path, bool grid) {   // :45-182
        V3 diffuse(1, 1, 1), ambient(0, 0, 0), specular(0, 0, 0), emission(0, 0, 0), atten(1, 0, 0);
        float shininess = 40;
        std::stack<M4> stk;
        stk.push(M4(1.f));
        std::ifstream f(path.c_str());
        opened = f.is_open();
        for (std::string line; std::getline(f, line);) {
            std::stringstream ls(line);
            std::string cmd;
            ls >> cmd;
            if (cmd == "size") ls >> cam.width >> cam.height;
            else if (cmd == "camera")
                ls >> cam.eye[0] >> cam.eye[1] >> cam.eye[2] >> cam.center[0] >> cam.center[1] >> cam.center[2]
                   >> cam.up[0] >> cam.up[1] >> cam.up[2] >> cam.fov;
            else if (cmd == "point" || cmd == "directional") {
                Lamp l;
                ls >> l.pos[0] >> l.pos[1] >> l.pos[2] >> l.color[0] >> l.color[1] >> l.color[2];
                if (cmd == "point") { l.kind = POINT_L; l.atten = atten; }
                else { l.kind = DIR_L; l.atten = V3(1, 0, 0); }   // reference leaves it uninitialised and never reads it
                lamps.push_back(l);
            } else if (cmd == "vertex") {
                V3 p;
                ls >> p[0] >> p[1] >> p[2];
                verts.push_back(p);
            } else if (cmd == "tri") {
                int a, b, c;
                ls >> a >> b >> c;
                Tri* t = new Tri;
                t->A = verts[a]; t->B = verts[b]; t->C = verts[c];
                t->mat = Mat{ambient, diffuse, specular, emission, shininess};
                t->xf = stk.top();
                t->id = (int)prims.size();
                prims.push_back(t);
            } else if (cmd == "sphere") {
                V3 c; float r;
                ls >> c[0] >> c[1] >> c[2] >> r;
                Ball* s = new Ball;
                s->S = c; s->r = r;
                s->mat = Mat{ambient, diffuse, specular, emission, shininess};
                s->xf = stk.top();
                s->id = (int)prims.size();
                prims.push_back(s);
            }
            else if (cmd == "ambient") ls >> ambient[0] >> ambient[1] >> ambient[2];
            else if (cmd == "specular") ls >> specular[0] >> specular[1] >> specular[2];
            else if (cmd == "diffuse") ls >> diffuse[0] >> diffuse[1] >> diffuse[2];
            else if (cmd == "emission") ls >> emission[0] >> emission[1] >> emission[2];
            else if (cmd == "shininess") ls >> shininess;
            else if (cmd == "attenuation") ls >> atten[0] >> atten[1] >> atten[2];
            else if (cmd == "pushTransform") stk.push(stk.top());
            else if (cmd == "popTransform") { if (stk.size() > 1) stk.pop(); }
            else if (cmd == "translate") { V3 v; ls >> v[0] >> v[1] >> v[2]; stk.top() = translate(stk.top(), v); }
            else if (cmd == "rotate") {
                V3 ax; float deg;
                ls >> ax[0] >> ax[1] >> ax[2] >> deg;
                stk.top() = stk.top() * rotate(M4(1.f), deg * PI_F / 180.f, ax);
            } else if (cmd == "scale") { V3 v; ls >> v[0] >> v[1] >> v[2]; stk.top() = stk.top() * scale(M4(1.f), v); }
            else if (cmd == "output") ls >> output;
            // anything else (comments, maxdepth, maxverts) is ignored, as in the reference
        }
        if (grid) { lattice = new Lattice(prims); accel.reset(lattice); }
        else {
            Bag* b = new Bag;
            b->items = prims;
            accel.reset(b);
        }
    }
};

// src/main.cpp:30-45
V3 light_term(V3 rayDir, V3 L, V3 N, const Mat* m, V3 lc) {
    CNT(light_evals, 1);
    const float lambert = clamp01(dot(normalize(L), N));
    V3 diff = m->diffuse * lc * lambert;
    V3 half = normalize(normalize(L) + normalize(-rayDir));
    const float ph = clamp01(dot(half, N));
    const float shine = std::pow(std::max(ph, 0.f), m->shininess);
    V3 spec = m->specular * lc * shine;
    return diff + spec;
}

// src/main.cpp:47-79
V3 shadow_term(WHit frag, V3 rayDir, Scene& sc) {
    V3 acc(0, 0, 0);
    const Mat* m = &sc.prims[frag.prim]->mat;
    for (const Lamp& l : sc.lamps) {
        if (l.kind == POINT_L) {
            V3 L = normalize(l.pos - frag.p);
            V3 So = frag.p + sc.bias * L;
            float tL = dot(l.pos - frag.p, L);
            CNT(shadow, 1);
            WHit s = sc.accel->hit(So, L, tL);
            if (!s.ok) {
                float att = l.atten[0] + l.atten[1] * tL + l.atten[2] * tL * tL;
                acc += light_term(rayDir, L, frag.n, m, l.color) / att;
            }
        } else if (l.kind == DIR_L) {
            V3 L = normalize(l.pos);
            V3 So = frag.p + sc.bias * L;
            CNT(shadow, 1);
            WHit s = sc.accel->hit(So, L);
            if (!s.ok) acc += light_term(rayDir, L, frag.n, m, l.color);
        }
    }
    return clamp_rgb(acc);
}

// src/main.cpp:81-107; first_prim receives the primary hit's primitive index (-1 = miss)
V3 trace(V3 O, V3 D, Scene& sc, const float depth, int* first_prim) {
    WHit frag = sc.accel->hit(O, normalize(D));
    if (first_prim) *first_prim = frag.ok ? frag.prim : -1;
    if (frag.ok) {
        V3 refl(0, 0, 0);
        if (depth > 0) {
            V3 N = frag.n;
            V3 V = normalize(-D);
            V3 R = (2 * dot(V, N) * N) - V;
            V3 P = frag.p + sc.bias * R;
            CNT(reflect, 1);
            refl = trace(P, R, sc, depth - 1, nullptr);
        }
        V3 sh = shadow_term(frag, D, sc);
        const Mat& m = sc.prims[frag.prim]->mat;
        return clamp_rgb(m.ambient + m.emission + sh + m.specular * refl);
    }
    return V3(0, 0, 0);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C entry points for ctypes (tests and bench.py's cpu_baseline leg only)
// ---------------------------------------------------------------------------------------------
extern "C" {

static int g_reference_schedule = 0;      // see ref_set_schedule

void* ref_load(const char* path, int use_grid) {
    Scene* s = new Scene;
    s->read(path, use_grid != 0);
    if (!s->opened) { delete s; return nullptr; }
    s->cam.axes();                                   // src/main.cpp:112
    return s;
}
void ref_free(void* h) { delete (Scene*)h; }

// info[0..]: width, height, n_prims, n_lights, n_verts
void ref_info(void* h, int* info) {
    Scene* s = (Scene*)h;
    info[0] = s->cam.width; info[1] = s->cam.height; info[2] = (int)s->prims.size();
    info[3] = (int)s->lamps.size(); info[4] = (int)s->verts.size();
}
const char* ref_output_name(void* h) { return ((Scene*)h)->output.c_str(); }

// camera: eye(3) center(3) up(3) fov u(3) v(3) w(3) = 19 floats
void ref_camera(void* h, float* out) {
    Cam& c = ((Scene*)h)->cam;
    float v[19] = {c.eye.x, c.eye.y, c.eye.z, c.center.x, c.center.y, c.center.z, c.up.x, c.up.y, c.up.z, c.fov,
                   c.u.x, c.u.y, c.u.z, c.v.x, c.v.y, c.v.z, c.w.x, c.w.y, c.w.z};
    memcpy(out, v, sizeof v);
}
// lights: per light kind(1) pos(3) color(3) atten(3) = 10 floats
void ref_lights(void* h, float* out) {
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->lamps.size(); i++) {
        const Lamp& l = s->lamps[i];
        float v[10] = {(float)l.kind, l.pos.x, l.pos.y, l.pos.z, l.color.x, l.color.y, l.color.z, l.atten.x, l.atten.y, l.atten.z};
        memcpy(out + 10 * i, v, sizeof v);
    }
}
// per primitive: kind (0 tri, 1 sphere); geom 9 floats (tri A,B,C / sphere S,r,0..); material 13 floats;
// transform 16 floats column-major; inverse 16 floats column-major; world box 6 floats (lo, hi)
void ref_prims(void* h, int* kind, float* geom, float* mat, float* xf, float* inv, float* box) {
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->prims.size(); i++) {
        Prim* p = s->prims[i];
        float* g = geom + 9 * i;
        if (Tri* t = dynamic_cast<Tri*>(p)) {
            kind[i] = 0;
            float v[9] = {t->A.x, t->A.y, t->A.z, t->B.x, t->B.y, t->B.z, t->C.x, t->C.y, t->C.z};
            memcpy(g, v, sizeof v);
        } else {
            Ball* b = static_cast<Ball*>(p);
            kind[i] = 1;
            float v[9] = {b->S.x, b->S.y, b->S.z, b->r, 0, 0, 0, 0, 0};
            memcpy(g, v, sizeof v);
        }
        const Mat& m = p->mat;
        float mv[13] = {m.ambient.x, m.ambient.y, m.ambient.z, m.diffuse.x, m.diffuse.y, m.diffuse.z,
                        m.specular.x, m.specular.y, m.specular.z, m.emission.x, m.emission.y, m.emission.z, m.shininess};
        memcpy(mat + 13 * i, mv, sizeof mv);
        M4 iv = inverse(p->xf);
        for (int j = 0; j < 4; j++)
            for (int r = 0; r < 4; r++) { xf[16 * i + 4 * j + r] = p->xf[j][r]; inv[16 * i + 4 * j + r] = iv[j][r]; }
        Box b = p->extent();
        float bv[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
        memcpy(box + 6 * i, bv, sizeof bv);
    }
}

// grid: bounds lo(3) hi(3); returns total number of cell references.  With ids == NULL only counts
// (cell_count[3375]) are filled; otherwise ids receives the concatenated per-cell lists in cell order.
long ref_grid(void* h, float* bounds, int* cell_count, int* ids) {
    Scene* s = (Scene*)h;
    Lattice* g = s->lattice;
    if (!g) return -1;
    float b[6] = {g->lo.x, g->lo.y, g->lo.z, g->hi.x, g->hi.y, g->hi.z};
    memcpy(bounds, b, sizeof b);
    long total = 0;
    int n = int(g->res.x) * int(g->res.y) * int(g->res.z);
    for (int c = 0; c < n; c++) {
        cell_count[c] = (int)g->cells[c].items.size();
        if (ids)
            for (Prim* p : g->cells[c].items) ids[total++] = p->id;
        else
            total += cell_count[c];
    }
    return total;
}

// one ray through the scene's accelerator: out = ok, t, p(3), n(3), prim  (9 floats)
void ref_intersect(void* h, const float* O, const float* D, float t_limit, float* out) {
    Scene* s = (Scene*)h;
    WHit w = s->accel->hit(V3(O[0], O[1], O[2]), V3(D[0], D[1], D[2]), t_limit);
    float v[9] = {w.ok ? 1.f : 0.f, w.t, w.p.x, w.p.y, w.p.z, w.n.x, w.n.y, w.n.z, (float)w.prim};
    memcpy(out, v, sizeof v);
}
// one primitive, object space (no transform applied): out = ok, t, n(3)
void ref_prim_hit(void* h, int prim, const float* O, const float* D, float* out) {
    Scene* s = (Scene*)h;
    OHit o = s->prims[prim]->hit(V3(O[0], O[1], O[2]), V3(D[0], D[1], D[2]));
    float v[5] = {o.ok ? 1.f : 0.f, o.t, o.n.x, o.n.y, o.n.z};
    memcpy(out, v, sizeof v);
}
// camera ray for pixel (x, y) at the given image size
void ref_camera_ray(void* h, int width, int height, int x, int y, float* out) {
    Scene* s = (Scene*)h;
    Cam c = s->cam;
    c.width = width; c.height = height;
    V3 d = c.ray(x, y);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

// The pixel loop (src/main.cpp:124-137): rows y = row_start, row_start + row_step, ... are rendered with
// OpenMP over rows; rgb / ids are FULL-image buffers (width*height*3 floats / width*height ints), rows not
// rendered are left untouched.  width/height <= 0 use the scene file's `size`.  Returns the seconds spent
// in the loop (same timed region as the reference).  counters (9 x uint64, may be NULL) is filled only
// in the REF_COUNT build: rays, tri_tests, sph_tests, cell_steps, light_evals, hits, primary, shadow, reflect.
double ref_render(void* h, int width, int height, int max_depth, int row_start, int row_step,
                  float* rgb, int* ids, uint64_t* counters) {
    Scene* s = (Scene*)h;
    if (g_reference_schedule) omp_set_schedule(omp_sched_static, 0);
    else omp_set_schedule(omp_sched_dynamic, 1);
    if (width > 0) s->cam.width = width;
    if (height > 0) s->cam.height = height;
    const int W = s->cam.width, H = s->cam.height;
    if (row_step < 1) row_step = 1;
    const int nrows = (H - row_start + row_step - 1) / row_step;
#ifdef REF_COUNT
    Counters total;
    memset(&total, 0, sizeof total);
#endif
    auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel
    {
#ifdef REF_COUNT
        memset(&tl_cnt, 0, sizeof tl_cnt);
#endif
#pragma omp for schedule(runtime)
        for (int r = 0; r < nrows; r++) {
            int y = row_start + r * row_step;
            for (int x = 0; x < W; x++) {
                V3 d = s->cam.ray(x, y);
                int first = -1;
                CNT(primary, 1);
                V3 c = trace(s->cam.eye, d, *s, (float)max_depth, &first);
                size_t o = (size_t)y * W + x;
                if (rgb) { rgb[3 * o] = c.x; rgb[3 * o + 1] = c.y; rgb[3 * o + 2] = c.z; }
                if (ids) ids[o] = first;
            }
        }
#ifdef REF_COUNT
#pragma omp critical
        {
            uint64_t* a = (uint64_t*)&total;
            uint64_t* b = (uint64_t*)&tl_cnt;
            for (size_t k = 0; k < sizeof(Counters) / 8; k++) a[k] += b[k];
        }
#endif
    }
    auto t1 = std::chrono::high_resolution_clock::now();
#ifdef REF_COUNT
    if (counters) memcpy(counters, &total, sizeof total);
#else
    if (counters) memset(counters, 0, 9 * sizeof(uint64_t));
#endif
    return std::chrono::duration<double>(t1 - t0).count();
}

// 8-bit conversion exactly as the reference's writers do it (src/Image3f.h:72-74,163-165): truncation.
void ref_quantize(const float* rgb, long n_values, unsigned char* out) {
    for (long i = 0; i < n_values; i++) out[i] = (unsigned char)(clamp01(rgb[i]) * 255);
}

// number of OpenMP threads ref_render will use
int ref_threads();

int ref_has_counters() {
#ifdef REF_COUNT
    return 1;
#else
    return 0;
#endif
}

int ref_threads() { return omp_get_max_threads(); }

// How ref_render hands the rows to the threads.  reference_schedule != 0: contiguous blocks of rows per thread, what the
// reference's plain `#pragma omp parallel for` does (src/main.cpp:126; the timed CPU baseline uses this); 0: one row
// at a time to whichever thread is free (the default here: same pixels, sooner, for the tests).
void ref_set_schedule(int reference_schedule) { g_reference_schedule = reference_schedule; }

}  // extern "C"
