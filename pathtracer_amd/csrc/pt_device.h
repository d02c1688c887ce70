// pt_device.h -- device-side arithmetic of the hot path (product code, gfx950).
//
// Every routine restates the fp32 expression trees of the reference function it names
// (compile with -ffp-contract=off; IEEE divide/sqrt are hipcc's default), but on the
// flattened scene of pt_scene.h: no virtual calls, no HitInfo copies, attribute
// interpolation only for the winning primitive.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_scene.h"

#define PT_DEV __device__ __forceinline__
// Explicit address spaces for the loads whose pointer the compiler cannot trace to its origin (a pointer read from a table in
// memory, or one of two pointers chosen at run time): without them such a load is a flat load.
#define PT_GLOBAL_AS __attribute__((address_space(1)))
#define PT_LDS_AS __attribute__((address_space(3)))
typedef unsigned int pt_u2v __attribute__((ext_vector_type(2)));
typedef float pt_f4v __attribute__((ext_vector_type(4)));

namespace ptd {

constexpr float kPI = 3.1415926535897932384626433f;       // reference include/constants.h:5-9
constexpr float kTwoPI = kPI * 2.0f;
constexpr float kInvPI = 1.0f / kPI;
constexpr float kBig = 1.0e30f;
constexpr float kBias = 0.0001f;
constexpr float kFltEps = 1.1920928955078125e-7f;
enum { SIDE_FRONT = 1, SIDE_BACK = 2, SIDE_BOTH = 3 };

// ------------------------------------------------------------------ transcendental functions
// The reference calls libm's float functions (sinf, cosf, powf, logf, acosf, ...).  glibc's are
// correctly rounded in all but a vanishing fraction of arguments; ROCm's fp32 device functions are
// 1-2 ulp.  Evaluating in fp64 and rounding once gives the correctly rounded fp32 result (up to
// double-rounding cases ~1e-8), so device samples follow the same branches as the CPU reference.
// These sit in shading code, far off the traversal critical path.
// sin and cos of one angle together.  Every call site's angle lies in [0, 2 pi] (2 pi * U, or an acos): one Cody-Waite
// step (k * pi/2 in two parts) and the fdlibm kernel polynomials (|error| < 2^-57) in fp64, rounded once to fp32 --
// a third of the instructions of two libm calls, which matters because shading is instruction-bound.
PT_DEV void a_sincosf(float xf, float& s, float& c) {
    const double x = (double)xf;
    if (!(xf >= -0.5f && xf <= 7.0f)) { s = (float)sin(x); c = (float)cos(x); return; }
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                        -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double sn = fma(r * z, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                        2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double ss = (q & 1) ? cs : sn, cc = (q & 1) ? sn : cs;
    s = (float)((q & 2) ? -ss : ss);
    c = (float)((q == 1 || q == 2) ? -cc : cc);
}
PT_DEV float a_sinf(float x) { return (float)sin((double)x); }
PT_DEV float a_cosf(float x) { return (float)cos((double)x); }
// powf as exp(y log x) in fp64: a relative error of about |y log x| 2^-52 <= 1e-14 against the float's half-ulp of 6e-8, i.e. the correctly
// rounded float but for one evaluation in ten million -- the class libm's own powf is in (0.8 ulp) -- at under half the instructions of the
// fp64 pow (k_resolve takes three per sample for the gamma curve, k_shade three per textured vertex for sRGB).  Every call site has a
// positive constant exponent or, in the clearcoat lobe, 1 - r2 >= 0 with a finite positive base: x = 0 -> 0, x < 0 -> NaN, y = 0 -> 1 as powf.
PT_DEV float a_powf(float x, float y) {
    if (y == 0.0f) return 1.0f;
    return (float)exp((double)y * log((double)x));
}
PT_DEV float a_logf(float x) { return (float)log((double)x); }
PT_DEV float a_acosf(float x) { return (float)acos((double)x); }
PT_DEV float a_asinf(float x) { return (float)asin((double)x); }
PT_DEV float a_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }

// ------------------------------------------------------------------ vectors (cyVector.h semantics)
struct F3 { float x, y, z; };
struct F2 { float x, y; };
PT_DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }
PT_DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV F3 operator/(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV F3 operator+(F3 a, float v) { return f3(a.x + v, a.y + v, a.z + v); }
PT_DEV F3 operator-(F3 a, float v) { return f3(a.x - v, a.y - v, a.z - v); }
PT_DEV F3 operator*(F3 a, float v) { return f3(a.x * v, a.y * v, a.z * v); }
PT_DEV F3 operator*(float v, F3 a) { return f3(a.x * v, a.y * v, a.z * v); }
PT_DEV F3 operator/(F3 a, float v) { return f3(a.x / v, a.y / v, a.z / v); }
PT_DEV F3 neg(F3 a) { return f3(-a.x, -a.y, -a.z); }
PT_DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV F3 cross(F3 a, F3 p) { return f3(a.y * p.z - a.z * p.y, a.z * p.x - a.x * p.z, a.x * p.y - a.y * p.x); }
PT_DEV float len2(F3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PT_DEV float len(F3 a) { return sqrtf(len2(a)); }
PT_DEV F3 normalized(F3 a) { return a / len(a); }
PT_DEV float max3(F3 a) { return a.x > a.y ? (a.x > a.z ? a.x : a.z) : (a.y > a.z ? a.y : a.z); }
PT_DEV float sum3(F3 a) { return a.x + a.y + a.z; }
PT_DEV float cymax(float a, float b) { return a >= b ? a : b; }
PT_DEV F2 f2(float x, float y) { F2 r; r.x = x; r.y = y; return r; }
PT_DEV F2 operator-(F2 a, F2 b) { return f2(a.x - b.x, a.y - b.y); }
PT_DEV float cross2(F2 a, F2 p) { return (-a.y) * p.x + a.x * p.y; }

// Matrix3 * Vec3 (cyMatrix.h:684-688), column-major cells
PT_DEV F3 m3mul(const float* c, F3 p) {
    return f3(p.x * c[0] + p.y * c[3] + p.z * c[6], p.x * c[1] + p.y * c[4] + p.z * c[7], p.x * c[2] + p.y * c[5] + p.z * c[8]);
}
// TransposeMult (Transformation.h:87-94)
PT_DEV F3 m3tmul(const float* c, F3 d) {
    return f3(c[0] * d.x + c[1] * d.y + c[2] * d.z, c[3] * d.x + c[4] * d.y + c[5] * d.z, c[6] * d.x + c[7] * d.y + c[8] * d.z);
}
// Matrix4 * Vec4(p, w) -> xyz (cyMatrix.h:1941-1957: (x*c0 + z*c8) + (y*c4 + w*c12))
PT_DEV F3 m4mulp(const float* c, F3 p, float w) {
    return f3((p.x * c[0] + p.z * c[8]) + (p.y * c[4] + w * c[12]),
              (p.x * c[1] + p.z * c[9]) + (p.y * c[5] + w * c[13]),
              (p.x * c[2] + p.z * c[10]) + (p.y * c[6] + w * c[14]));
}

// ------------------------------------------------------------------ RNG (specification in DESIGN.md)
PT_DEV uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
struct Rng {
    uint32_t key;      // hash of (seed, pixel, sample)
    uint32_t draw;
    PT_DEV void init(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t d) {
        uint32_t h = fmix32(seed + 0x9E3779B9u);
        h = fmix32(h ^ (pixel + 0x7F4A7C15u));
        key = fmix32(h ^ (sample * 0x85EBCA77u + 0xC2B2AE3Du));
        draw = d;
    }
    PT_DEV int next() { uint32_t h = fmix32(key ^ (draw * 0x27D4EB2Fu + 0x165667B1u)); draw++; return (int)(h >> 1); }
    PT_DEV float U() { return (float)next() / 2147483648.0f; }       // rand() / (float)RAND_MAX
};
constexpr uint32_t kPixelOffsetSample = 0xFFFFFFFFu;

// ------------------------------------------------------------------ rays / camera
struct Ray { F3 p, d; };

// src/utils.cpp:67-79 + src/raytracer.cpp:272-316 + src/sampler.cpp:15-32
PT_DEV float halton(int index, int base) {      // include/scene.h:62-71
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) { r += f * (i % base); f /= (float)base; }
    return r;
}
// Camera ray context of (pixel, sample).  jitter == false gives GenCameraRayContext(x, y, 0, 0).
PT_DEV void camera_rays(const DCamera& c, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t width, bool jitter,
                        Ray& mainRay, F3& rightDir, F3& topDir) {
    const int y = (int)(pixel / width), x = (int)(pixel - (uint32_t)y * width);
    float finalX = 0.0f, finalY = 0.0f;
    if (jitter) {
        Rng po; po.init(seed, pixel, kPixelOffsetSample, 0);      // PixelContext ctor draws (pathtracer.h:15-19)
        const float offx = po.U(), offy = po.U();
        finalX = (halton((int)sample, 2) - 0.5f) + offx;
        if (finalX >= 0.5f) finalX -= 1.0f;
        finalY = (halton((int)sample, 3) - 0.5f) + offy;
        if (finalY >= 0.5f) finalY -= 1.0f;
    }
    Rng rng; rng.init(seed, pixel, sample, 0);
    const float S = rng.U();
    const float r = sqrtf(S) * c.dof;
    const float theta = rng.U() * kTwoPI;
    float sin_t, cos_t;
    a_sincosf(theta, sin_t, cos_t);
    const float lx = r * cos_t, ly = r * sin_t;
    const F3 right = ld3(c.right), up = ld3(c.up), front = ld3(c.front);
    const F3 lens = right * lx + up * ly;
    mainRay.p = ld3(c.pos) + right * lx + up * ly;
    F3 dir = (right * (-1.0f * c.plane_w * 0.5f + (float)x * c.texel_w + 0.5f * c.texel_w + finalX * c.texel_w) +
              up * (c.plane_h * 0.5f - (float)y * c.texel_h - 0.5f * c.texel_h - finalY * c.texel_h) +
              front * c.focaldist) - lens;
    rightDir = normalized(dir + right * c.texel_w * 1.0f);
    topDir = normalized(dir + up * c.texel_h * 1.0f);
    mainRay.d = dir / len(dir);
}

// ------------------------------------------------------------------ node transforms (node.h:174-201)
PT_DEV Ray to_node(const DNodeXf& n, const Ray& r) {
    const F3 pos = ld3(n.pos);
    Ray o;
    o.p = m3mul(n.itm, r.p - pos);
    o.d = m3mul(n.itm, (r.p + r.d) - pos) - o.p;
    return o;
}
PT_DEV Ray to_inst(const DScene& S, const DInst& in, Ray r) {
    for (int i = 0; i < in.chain_len; i++) r = to_node(S.nodes[in.chain[i]], r);
    return r;
}

// Ray into the space of a walk record's object (pt_scene.h): ToNodeCoords level by level, root first.  Every chain
// starts at the scene's root node, so callers that walk several records for one ray apply level 0 once
// (first_level = 1 with the ray already in the root's space): same arithmetic, done once instead of per object.
PT_DEV Ray rec_ray(const DScene& S, const float4* R, uint32_t chain_len, uint32_t inst, Ray r, uint32_t first_level = 0) {
    for (uint32_t l = first_level; l < chain_len; l++) {
        if (l < (uint32_t)kRecLevels) {
            const float4 a = R[7 + 3 * l], b = R[8 + 3 * l], c = R[9 + 3 * l];
            const float itm[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x};
            const F3 pos = f3(c.y, c.z, c.w);
            Ray o;
            o.p = m3mul(itm, r.p - pos);
            o.d = m3mul(itm, (r.p + r.d) - pos) - o.p;
            r = o;
        } else r = to_node(S.nodes[S.insts[inst].chain[l]], r);
    }
    return r;
}

// ------------------------------------------------------------------ slab test (bvh.h:88-129, scene.cpp:3-40)
// inv = 1/dir is computed once per (ray, instance): same values as the reference's per-call division.
PT_DEV bool slab(const float* b, F3 p, F3 inv, float& tnear) {
    const bool s0 = inv.x < 0, s1 = inv.y < 0, s2 = inv.z < 0;
    float tmin = ((s0 ? b[3] : b[0]) - p.x) * inv.x;
    float tmax = ((s0 ? b[0] : b[3]) - p.x) * inv.x;
    const float tymin = ((s1 ? b[4] : b[1]) - p.y) * inv.y;
    const float tymax = ((s1 ? b[1] : b[4]) - p.y) * inv.y;
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    const float tzmin = ((s2 ? b[5] : b[2]) - p.z) * inv.z;
    const float tzmax = ((s2 ? b[2] : b[5]) - p.z) * inv.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    if (tzmin > tmin) tmin = tzmin;
    tnear = tmin;
    return true;
}

// The same test without early exits: every operand is used unconditionally, so the compiler keeps the box loads together
// at the top instead of sinking them into the exits (a dependent round trip each).  Same comparisons, same results.
PT_DEV bool slab_straight(const float* b, F3 p, F3 inv, float& tnear) {
    const bool s0 = inv.x < 0, s1 = inv.y < 0, s2 = inv.z < 0;
    float tmin = ((s0 ? b[3] : b[0]) - p.x) * inv.x;
    float tmax = ((s0 ? b[0] : b[3]) - p.x) * inv.x;
    const float tymin = ((s1 ? b[4] : b[1]) - p.y) * inv.y;
    const float tymax = ((s1 ? b[1] : b[4]) - p.y) * inv.y;
    bool ok = !((tmin > tymax) || (tymin > tmax));
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    const float tzmin = ((s2 ? b[5] : b[2]) - p.z) * inv.z;
    const float tzmax = ((s2 ? b[2] : b[5]) - p.z) * inv.z;
    ok = ok && !((tmin > tzmax) || (tzmin > tmax));
    if (tzmin > tmin) tmin = tzmin;
    tnear = tmin;
    return ok;
}

// Both child boxes of a 64-B node record (w0.xyzw w1.xy = left min/max, w1.zw w2.xyzw = right min/max), the reference's
// literal form: sign-selected planes, sequential interval tests, NaN comparisons fall through as "hit" (bvh.h:95-129).
PT_DEV void slab_pair_exact(float4 w0, float4 w1, float4 w2, F3 o, F3 inv, bool& hl, float& tl, bool& hr, float& tr) {
    const bool s0 = inv.x < 0, s1 = inv.y < 0, s2 = inv.z < 0;
    {
        float tmin = ((s0 ? w0.w : w0.x) - o.x) * inv.x, tmax = ((s0 ? w0.x : w0.w) - o.x) * inv.x;
        const float tymin = ((s1 ? w1.x : w0.y) - o.y) * inv.y, tymax = ((s1 ? w0.y : w1.x) - o.y) * inv.y;
        bool ok = !((tmin > tymax) || (tymin > tmax));
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        const float tzmin = ((s2 ? w1.y : w0.z) - o.z) * inv.z, tzmax = ((s2 ? w0.z : w1.y) - o.z) * inv.z;
        ok = ok && !((tmin > tzmax) || (tzmin > tmax));
        if (tzmin > tmin) tmin = tzmin;
        hl = ok; tl = tmin;
    }
    {
        float tmin = ((s0 ? w2.y : w1.z) - o.x) * inv.x, tmax = ((s0 ? w1.z : w2.y) - o.x) * inv.x;
        const float tymin = ((s1 ? w2.z : w1.w) - o.y) * inv.y, tymax = ((s1 ? w1.w : w2.z) - o.y) * inv.y;
        bool ok = !((tmin > tymax) || (tymin > tmax));
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        const float tzmin = ((s2 ? w2.w : w2.x) - o.z) * inv.z, tzmax = ((s2 ? w2.x : w2.w) - o.z) * inv.z;
        ok = ok && !((tmin > tzmax) || (tzmin > tmax));
        if (tzmin > tmin) tmin = tzmin;
        hr = ok; tr = tmin;
    }
}
// The same test for a ray whose 1/d components are finite and non-zero and whose origin is finite (slab_fast_ok), without the
// sign selects and the sequential tests.  For such a ray no product below is NaN (box planes are finite, or all NaN in a chain
// record: then "not hit", which the caller treats like the literal form's "hit, never beyond" -- chain records have no internal
// children).  Rounding is monotone, so for 1/d > 0 the lower plane gives the smaller product and for 1/d < 0 the larger: min / max
// of the two products ARE the reference's sign-selected tmin / tmax, bit for bit.  Each axis interval is then non-empty, three
// non-empty intervals have a common point iff they meet pairwise (the reference's two-stage test), i.e. iff
// max(mins) <= min(maxs), and the reference's final tmin is max(mins) (equal up to the sign of a zero, which no use of it sees).
PT_DEV void slab_one_fast(float x0, float y0, float z0, float x1, float y1, float z1, F3 o, F3 inv, bool& hit, float& tn) {
    const float ax = (x0 - o.x) * inv.x, bx = (x1 - o.x) * inv.x;
    const float ay = (y0 - o.y) * inv.y, by = (y1 - o.y) * inv.y;
    const float az = (z0 - o.z) * inv.z, bz = (z1 - o.z) * inv.z;
    const float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    hit = tnear <= tfar; tn = tnear;
}
PT_DEV void slab_pair_fast(float4 w0, float4 w1, float4 w2, F3 o, F3 inv, bool& hl, float& tl, bool& hr, float& tr) {
    slab_one_fast(w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, o, inv, hl, tl);
    slab_one_fast(w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, o, inv, hr, tr);
}
PT_DEV bool slab_fast_ok(F3 o, F3 inv) {
    const float big = __uint_as_float(0x7F800000u);
    return fabsf(inv.x) < big && fabsf(inv.y) < big && fabsf(inv.z) < big && inv.x != 0.0f && inv.y != 0.0f && inv.z != 0.0f &&
           fabsf(o.x) < big && fabsf(o.y) < big && fabsf(o.z) < big;
}

// Pruning guard.  The reference has no distance culling at all (bvh.h:88-129 ignores t); skipping a
// subtree whose slab entry distance lies beyond the best hit is exact in real arithmetic, but the
// triangle's t (plane equation, cancellation ~ eps * (|v0| + |o|) / |cos|) and the slab's entry distance round
// differently, so the cull keeps a margin: a subtree is skipped only when
// tnear > (bt + E) * (1 + 1e-4) + 1e-4 * |o|_1, E = the largest error bound of the candidates accepted so far
// (tri_closest_key).  What this covers and what it cannot (a culled triangle met at under ~0.3 degrees whose
// neighbours were not): DESIGN.md "Distance cull".  MI355PT_NO_CULL=1 removes the cull for verification.
PT_DEV float prune_abs(F3 o) { return 1e-4f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)); }
PT_DEV bool beyond(float tnear, float bt, float pabs) { return tnear > bt * 1.0001f + pabs; }

// ------------------------------------------------------------------ triangle (model.h:227-347)
struct Tri { F3 v0, v1, v2; uint32_t face; };
PT_DEV Tri load_tri(const float4* tris, uint32_t slot) {
    const float4 a = tris[3 * (size_t)slot], b = tris[3 * (size_t)slot + 1], c = tris[3 * (size_t)slot + 2];
    Tri t;
    t.v0 = f3(a.x, a.y, a.z); t.v1 = f3(a.w, b.x, b.y); t.v2 = f3(b.z, b.w, c.x);
    t.face = __float_as_uint(c.y);
    return t;
}
// glm arithmetic (func_geometric.inl:48-54,68-79; func_matrix.inl:269-291; type_mat3x3.inl:467-474)
PT_DEV float gdot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV F3 gcross(F3 x, F3 y) { return f3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
PT_DEV F3 gnormalize(F3 v) { return v * (1.0f / sqrtf(gdot(v, v))); }

// Barycentric part of IntersectRayWithFace for point P and matrix columns (c0, v01, v02);
// GLM = true uses glm::inverse (one reciprocal), false uses cy Matrix3::Invert (divide by det).
template <bool GLM>
PT_DEV void tri_areas(F3 c0, F3 v01, F3 v02, F3 v0, F3 v1, F3 v2, F3 P, float& a0, float& a1, float& a2, float& a) {
    // m[col][row]: m0 = c0, m1 = v01, m2 = v02.  Only rows y,z of the inverse are used.
    float i01, i11, i21, i02, i12, i22;      // inverse[col][row] for rows 1,2
    if (GLM) {
        const float ood = 1.0f / (+c0.x * (v01.y * v02.z - v02.y * v01.z) - v01.x * (c0.y * v02.z - v02.y * c0.z) + v02.x * (c0.y * v01.z - v01.y * c0.z));
        i01 = -(c0.y * v02.z - v02.y * c0.z) * ood;
        i11 = +(c0.x * v02.z - v02.x * c0.z) * ood;
        i21 = -(c0.x * v02.y - v02.x * c0.y) * ood;
        i02 = +(c0.y * v01.z - v01.y * c0.z) * ood;
        i12 = -(c0.x * v01.z - v01.x * c0.z) * ood;
        i22 = +(c0.x * v01.y - v01.x * c0.y) * ood;
    } else {
        // cyMatrix.h:799-820 with cell[0..2]=c0, [3..5]=v01, [6..8]=v02; inverse cell[1],[4],[7] = row 1, [2],[5],[8] = row 2
        const float n0 = (v01.y * v02.z - v01.z * v02.y);
        const float n3 = (v01.z * v02.x - v01.x * v02.z);
        const float n6 = (v01.x * v02.y - v01.y * v02.x);
        const float det = c0.x * n0 + c0.y * n3 + c0.z * n6;
        i01 = (c0.z * v02.y - c0.y * v02.z) / det;
        i11 = (c0.x * v02.z - c0.z * v02.x) / det;
        i21 = (c0.y * v02.x - c0.x * v02.y) / det;
        i02 = (c0.y * v01.z - c0.z * v01.y) / det;
        i12 = (c0.z * v01.x - c0.x * v01.z) / det;
        i22 = (c0.x * v01.y - c0.y * v01.x) / det;
    }
    // rows y,z of M^-1 * v.  glm: m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z; cy: p.x*cell[r] + p.y*cell[3+r] + p.z*cell[6+r]
    // (same products, same order; cy multiplies v.x*cell, glm cell*v.x: commutative)
#define PT_ROW1(v) (i01 * (v).x + i11 * (v).y + i21 * (v).z)
#define PT_ROW2(v) (i02 * (v).x + i12 * (v).y + i22 * (v).z)
    const F2 q0 = f2(PT_ROW1(v0), PT_ROW2(v0)), q1 = f2(PT_ROW1(v1), PT_ROW2(v1)), q2 = f2(PT_ROW1(v2), PT_ROW2(v2)), qp = f2(PT_ROW1(P), PT_ROW2(P));
#undef PT_ROW1
#undef PT_ROW2
    const F2 pv0 = q0 - qp, pv1 = q1 - qp, pv2 = q2 - qp;
    a0 = cross2(pv1, pv2); a1 = cross2(pv2, pv0); a2 = cross2(pv0, pv1);
    a = cross2(q1 - q0, q2 - q0);
}

// Closest-hit candidate test.  (bt, bslot) is the current best in (t, slot) order; the slot order
// IS the reference's tie-break order (bvh_build.h).  Returns true when the candidate replaces it.
// The early exit before the matrix inverse cannot change the outcome: a candidate that is not
// better in (t, slot) would be discarded anyway.
PT_DEV bool tri_closest(const Tri& T, F3 o, F3 d, int side, uint32_t slot, float& bt, uint32_t& bslot) {
    const F3 v01 = T.v1 - T.v0, v02 = T.v2 - T.v0;
    const F3 n = gcross(v01, v02);
    const float dn = gdot(d, n);
    if (fabsf(dn) <= kFltEps) return false;
    const float t = (gdot(T.v0, n) - gdot(o, n)) / dn;
    if (t < 0.0f) return false;
    const bool front = dn < 0.0f;
    if (side == SIDE_FRONT && !front) return false;
    if (side == SIDE_BACK && front) return false;
    if (!(t < bt || (t == bt && slot < bslot))) return false;
    const F3 P = o + d * t;
    float a0, a1, a2, a;
    tri_areas<true>(neg(d), v01, v02, T.v0, T.v1, T.v2, P, a0, a1, a2, a);
    if (a0 <= 0.0f) return false;
    if (a1 <= 0.0f) return false;
    if (a2 <= 0.0f) return false;
    bt = t; bslot = slot;
    return true;
}
// The same test against a packed (t, slot) key (see hit_key in mi355pt.hip): returns true, and the candidate's key,
// when the candidate is inside the triangle and strictly better than `cur`.
// `cerr` (set on acceptance): a bound on how far the computed distance can lie from the geometric one -- the two dot
// products of the numerator cancel, so their rounding (and the rounding of n itself) is divided by |d . n|:
// 2^-20 * (sum |v0_i n_i| + sum |o_i n_i|) / |d . n|  (about 16 half-ulps of the operands; a few ulp for an ordinary hit,
// the whole distance for a ray that grazes the triangle's plane).
PT_DEV bool tri_closest_key(const Tri& T, F3 o, F3 d, uint32_t slot, unsigned long long cur, unsigned long long& ck, float& cerr) {
    const F3 v01 = T.v1 - T.v0, v02 = T.v2 - T.v0;
    const F3 n = gcross(v01, v02);
    const float dn = gdot(d, n);
    if (fabsf(dn) <= kFltEps) return false;
    const float t = (gdot(T.v0, n) - gdot(o, n)) / dn;
    if (t < 0.0f) return false;
    const uint32_t tb = __float_as_uint(t);
    ck = ((unsigned long long)((t == 0.0f) ? 0u : tb) << 32) | (unsigned long long)((slot << 1) | (tb == 0x80000000u ? 1u : 0u));
    if (!(ck < cur)) return false;                 // also rejects a NaN distance, as the reference's `z < best.z` does
    const F3 P = o + d * t;
    float a0, a1, a2, a;
    tri_areas<true>(neg(d), v01, v02, T.v0, T.v1, T.v2, P, a0, a1, a2, a);
    if (a0 <= 0.0f) return false;
    if (a1 <= 0.0f) return false;
    if (a2 <= 0.0f) return false;
    cerr = 9.5367431640625e-7f * ((fabsf(T.v0.x * n.x) + fabsf(T.v0.y * n.y) + fabsf(T.v0.z * n.z)) + (fabsf(o.x * n.x) + fabsf(o.y * n.y) + fabsf(o.z * n.z))) / fabsf(dn);
    return true;
}
// Any-hit form for visibility (raytracer.cpp:99-130): front faces only, accepted iff t < bound.
PT_DEV bool tri_any_front(const Tri& T, F3 o, F3 d, float bound) {
    const F3 v01 = T.v1 - T.v0, v02 = T.v2 - T.v0;
    const F3 n = gcross(v01, v02);
    const float dn = gdot(d, n);
    if (fabsf(dn) <= kFltEps) return false;
    const float t = (gdot(T.v0, n) - gdot(o, n)) / dn;
    if (t < 0.0f) return false;
    if (!(dn < 0.0f)) return false;
    if (!(t < bound)) return false;
    const F3 P = o + d * t;
    float a0, a1, a2, a;
    tri_areas<true>(neg(d), v01, v02, T.v0, T.v1, T.v2, P, a0, a1, a2, a);
    return a0 > 0.0f && a1 > 0.0f && a2 > 0.0f;
}

// ------------------------------------------------------------------ BVH encoding (bvh_build.h)
constexpr int kStackSize = 48;        // stack entries per lane; the host checks tree depth + 4 <= kStackSize
constexpr int kTravBlock = 256;       // threads per block of the streaming kernels
constexpr uint32_t kLeafFlag = 0x80000000u, kLeafSlotMask = 0x07FFFFFFu;   // leaf item: flag | (count - 1) << 27 | first slot
constexpr uint32_t kInternal = 0xFFFFFFFFu, kChain = 0xFFFFFFFEu;         // count words of a node record
struct WorkCounters { uint32_t nodes, tris, top; };
struct HitRec { float t; int inst; int mesh; uint32_t slot; };     // closest-hit record of one path

// ------------------------------------------------------------------ plane (objects.cpp:560-614)
PT_DEV bool plane_hit(F3 o, F3 d, int side, float& t, bool& front, F3& P) {
    const float dn = d.z;                       // dir . (0,0,1)
    if (fabsf(dn - 0.0f) < 0.000001f) return false;
    const float tt = -1.0f * (o.z / dn);
    if (tt <= 0.0f) return false;
    const F3 p = o + d * tt;
    if (fabsf(p.x) >= 1.0f) return false;
    if (fabsf(p.y) >= 1.0f) return false;
    const bool f = dn < 0.0f;
    if (side == SIDE_FRONT && !f) return false;
    if (side == SIDE_BACK && f) return false;
    t = tt; front = f; P = p;
    return true;
}

// ------------------------------------------------------------------ sphere (objects.cpp:684-790)
// Unit sphere in object space.  Two roots: the near one is a front hit; when it lies behind the origin the far one
// is a back hit; a double root (delta == 0) is a front hit.  Same expression trees as the reference.
PT_DEV bool sphere_hit(F3 o, F3 d, int side, float& t, bool& front, F3& P) {
    const float a = len2(d);
    const float b = 2 * dot(o, d);
    const float c = len2(o) - 1;
    const float delta = b * b - 4 * a * c;
    if (delta < 0) return false;
    if (delta == 0) {
        const float dist = -1.0f * b / (2.0f * a);
        if (dist < 0.0f) return false;
        if (!(side & SIDE_FRONT)) return false;
        t = dist; front = true; P = o + d * dist;
        return true;
    }
    const float dist1 = (-1.0f * b - sqrtf(delta)) / (2.0f * a);
    const float dist2 = (-1.0f * b + sqrtf(delta)) / (2.0f * a);
    if (dist1 < 0.0f) {
        if (dist2 < 0.0f) return false;
        if (!(side & SIDE_BACK)) return false;
        t = dist2; front = false; P = o + d * dist2;
        return true;
    }
    if (!(side & SIDE_FRONT)) return false;
    t = dist1; front = true; P = o + d * dist1;
    return true;
}

}  // namespace ptd
