// pt_shade.h -- device-side shading arithmetic: hit reconstruction, textures, Disney and
// Cook-Torrance materials, rectangular area light, visibility (product code, gfx950).
// Reference functions restated are cited at each routine.
#pragma once
#include "pt_device.h"

namespace ptd {

// ------------------------------------------------------------------ surface hit (HitInfo, hitinfo.h:23-62)
struct Surf {
    float z;
    F3 p, N, T, B, uvw, du0, du1;
    int node, mtl, light;
    bool front;
};

// IntersectRayWithFace for the winning triangle only (model.h:227-347 attributes, :90-225 differentials)
PT_DEV F3 diff_tex(const Ray& dr, F3 pD, const Tri& T, F3 v01, F3 v02, F3 t0, F3 t1, F3 t2) {
    float a0, a1, a2, a;
    tri_areas<false>(neg(dr.d), v01, v02, T.v0, T.v1, T.v2, pD, a0, a1, a2, a);
    const float b0 = a0 / a, b1 = a1 / a, b2 = a2 / a;
    return b0 * t0 + b1 * t1 + b2 * t2;
}
// `needs_uv` / `with_tb`: the interpolated uvw (and its differentials) feed textures only, Tangent / Bitangent the
// normal map only.  Normals and uvw are read from the per-slot attribute record, in the same load phase as the triangle.
// The triangle and its normals are loaded by the caller before it walks the node chain (TriPre): they depend on the hit
// record only, and the walk's own dependent loads would otherwise sit in front of them.
struct TriPre { float4 t0, t1, t2, A0, A1, A2; };
PT_DEV TriPre load_tri_pre(const DScene& S, uint32_t slot) {
    TriPre p;
    const float4* t = S.tris + 3 * (size_t)slot; const float4* A = S.attrs + 5 * (size_t)slot;
    p.t0 = t[0]; p.t1 = t[1]; p.t2 = t[2]; p.A0 = A[0]; p.A1 = A[1]; p.A2 = A[2];
    return p;
}
PT_DEV void surf_triangle(const DScene& S, const DMesh* M, uint32_t slot, const TriPre& pre, const Ray& r, bool hasDiff, bool needs_uv, bool with_tb,
                          const Ray& rr, const Ray& tr, Surf& h) {
    Tri T;
    T.v0 = f3(pre.t0.x, pre.t0.y, pre.t0.z); T.v1 = f3(pre.t0.w, pre.t1.x, pre.t1.y); T.v2 = f3(pre.t1.z, pre.t1.w, pre.t2.x);
    T.face = __float_as_uint(pre.t2.y);
    const float4* A = S.attrs + 5 * (size_t)slot;
    const float4 A0 = pre.A0, A1 = pre.A1, A2 = pre.A2;
    float4 A3 = make_float4(0, 0, 0, 0), A4 = make_float4(0, 0, 0, 0);
    if (needs_uv) { A3 = A[3]; A4 = A[4]; }
    const F3 v01 = T.v1 - T.v0, v02 = T.v2 - T.v0;
    const F3 n = gcross(v01, v02);
    const float dn = gdot(r.d, n);
    const float t = (gdot(T.v0, n) - gdot(r.p, n)) / dn;
    const bool front = dn < 0.0f;
    const F3 P = r.p + r.d * t;
    float a0, a1, a2, a;
    tri_areas<true>(neg(r.d), v01, v02, T.v0, T.v1, T.v2, P, a0, a1, a2, a);
    const float b0 = a0 / a, b1 = a1 / a, b2 = a2 / a;
    const F3 n0 = f3(A0.x, A0.y, A0.z), n1 = f3(A0.w, A1.x, A1.y), n2 = f3(A1.z, A1.w, A2.x);
    F3 normal = gnormalize(b0 * n0 + b1 * n1 + b2 * n2);
    const float ns = normal.x + normal.y + normal.z;
    if (isnan(ns) || isinf(ns)) normal = gnormalize(n);
    const F3 t0 = f3(A2.y, A2.z, A2.w), t1 = f3(A3.x, A3.y, A3.z), t2 = f3(A3.w, A4.x, A4.y);
    h.T = f3(0, 0, 0); h.B = f3(0, 0, 0);
    if (with_tb && M->has_tangents) {
        const size_t fi = (size_t)M->face_base + S.faces[4 * (size_t)(M->face_base + T.face) + 3];      // face.indices[3], model.h:330-335
        h.T = ld3(S.tangents + 3 * fi); h.B = ld3(S.bitangents + 3 * fi);
    }
    h.p = P; h.z = t; h.front = front;
    h.N = front ? normal : -1.0f * normal;
    h.uvw = b0 * t0 + b1 * t1 + b2 * t2;
    if (hasDiff) {
        const F3 N = normalized(h.N);
        const float tR = dot(rr.p - h.p, N) / dot(rr.d, N);
        const F3 pR = rr.p + tR * rr.d;
        const float tT = dot(tr.p - h.p, N) / dot(tr.d, N);
        const F3 pT = tr.p + tT * tr.d;
        const F3 texR = diff_tex(rr, pR, T, v01, v02, t0, t1, t2);
        const F3 texT = diff_tex(tr, pT, T, v01, v02, t0, t1, t2);
        h.du0 = (texR - h.uvw) / 1.0f;
        h.du1 = (texT - h.uvw) / 1.0f;
    } else { h.du0 = f3(0, 0, 0); h.du1 = f3(0, 0, 0); }
}
// Plane::IntersectRay (objects.cpp:505-614)
PT_DEV F3 plane_tex(F3 p) { return p * 0.5f + f3(0.5f, 0.5f, 0.0f); }
PT_DEV void surf_plane(const Ray& r, bool hasDiff, const Ray& rr, const Ray& tr, Surf& h) {
    float t; bool front; F3 P;
    plane_hit(r.p, r.d, SIDE_BOTH, t, front, P);
    const F3 n = f3(0.0f, 0.0f, 1.0f);
    h.front = front; h.N = front ? n : -1.0f * n; h.p = P; h.z = t;
    h.T = f3(1.0f, 0.0f, 0.0f); h.B = f3(0.0f, 1.0f, 0.0f);
    h.uvw = plane_tex(P);
    if (hasDiff) {
        const F3 N = normalized(h.N);
        const float tR = -1.0f * dot(rr.p, N) / dot(rr.d, N);
        const F3 pR = rr.p + tR * rr.d;
        const float tT = -1.0f * dot(tr.p, N) / dot(tr.d, N);
        const F3 pT = tr.p + tT * tr.d;
        h.du0 = (plane_tex(pR) - h.uvw) / 1.0f;
        h.du1 = (plane_tex(pT) - h.uvw) / 1.0f;
    } else { h.du0 = f3(0, 0, 0); h.du1 = f3(0, 0, 0); }
}
// Sphere::IntersectRay (objects.cpp:620-790): N = the hit point (un-normalised, negated on the back side), Tangent and
// Bitangent stay at HitInfo::Init's zero, uvw = latitude / longitude of the hit point
PT_DEV F3 sphere_tex(F3 p, float reverseLength) {
    return f3(0.5f - a_atan2f(p.x, p.y) / (2.0f * kPI), 0.5f + a_asinf(p.z * reverseLength) / kPI, 0.0f);
}
PT_DEV void surf_sphere(const Ray& r, bool hasDiff, const Ray& rr, const Ray& tr, Surf& h) {
    float t; bool front; F3 P;
    sphere_hit(r.p, r.d, SIDE_BOTH, t, front, P);
    h.front = front; h.N = front ? P : -1.0f * P; h.p = P; h.z = t;
    h.T = f3(0, 0, 0); h.B = f3(0, 0, 0);
    h.uvw = sphere_tex(P, 1.0f);
    if (hasDiff) {
        const F3 N = normalized(h.N);
        const float tR = -1.0f * dot(rr.p - h.p, N) / dot(rr.d, N);
        const F3 pR = rr.p + tR * rr.d;
        const float tT = -1.0f * dot(tr.p - h.p, N) / dot(tr.d, N);
        const F3 pT = tr.p + tT * tr.d;
        h.du0 = (sphere_tex(pR, 1.0f / len(pR)) - h.uvw) / 1.0f;
        h.du1 = (sphere_tex(pT, 1.0f / len(pT)) - h.uvw) / 1.0f;
    } else { h.du0 = f3(0, 0, 0); h.du1 = f3(0, 0, 0); }
}
// FromNodeCoords up the chain (node.h:195-201), as TraceNode applies it level by level
// Tangent / Bitangent are read by normal mapping only (disneyMaterial.h:113-121): with_tb = false leaves them alone.
PT_DEV void surf_to_world(const DScene& S, const DInst& in, Surf& h, bool with_tb) {
    for (int i = in.chain_len - 1; i >= 0; i--) {
        const DNodeXf& n = S.nodes[in.chain[i]];
        h.p = m3mul(n.tm, h.p) + ld3(n.pos);
        h.N = normalized(m3tmul(n.itm, h.N));
        if (with_tb) {
            h.T = normalized(m3tmul(n.itm, h.T));
            h.B = normalized(m3tmul(n.itm, h.B));
        }
    }
}
// Full HitInfo of a recorded closest hit.
// Values nothing reads are not computed: the uv differentials feed the texture filter only, Tangent / Bitangent the
// normal map only, so a hit on a material without maps skips them (what IS computed is computed as the reference does).
PT_DEV void surf_from_hit(const DScene& S, const HitRec& hr, const Ray& wr, bool hasDiff, F3 rightDir, F3 topDir, Surf& h) {
    TriPre pre;
    pre.t0 = pre.t1 = pre.t2 = pre.A0 = pre.A1 = pre.A2 = make_float4(0, 0, 0, 0);
    if (hr.mesh >= 0) pre = load_tri_pre(S, hr.slot);
    const DInst& in = S.insts[hr.inst];
    h.node = in.node;
    h.mtl = in.material; h.light = in.light;
    const mi355pt_material& mt = S.materials[h.mtl];
    const bool with_tb = mt.normal_map >= 0;
    const bool needs_uv = with_tb || mt.albedo_map >= 0 || mt.roughness_map >= 0 || mt.metalness_map >= 0;
    hasDiff = hasDiff && needs_uv;
    Ray r = wr, rr, tr;
    rr.p = wr.p; rr.d = rightDir; tr.p = wr.p; tr.d = topDir;
    for (int i = 0; i < in.chain_len; i++) {
        const DNodeXf& n = S.nodes[in.chain[i]];
        r = to_node(n, r);
        if (hasDiff) { rr = to_node(n, rr); tr = to_node(n, tr); }
    }
    if (in.kind == MI355PT_OBJ_PLANE) surf_plane(r, hasDiff, rr, tr, h);
    else if (in.kind == MI355PT_OBJ_SPHERE) surf_sphere(r, hasDiff, rr, tr, h);
    else surf_triangle(S, with_tb ? &S.meshes[S.models[in.model].first_mesh + hr.mesh] : nullptr, hr.slot, pre, r, hasDiff, needs_uv, with_tb, rr, tr, h);
    surf_to_world(S, in, h, with_tb);
}

// ------------------------------------------------------------------ textures (scene.h:157-309, texture.cpp:102-150)
PT_DEV F3 tile_clamp(F3 uvw) {
    F3 u;
    u.x = uvw.x - (int)uvw.x; u.y = uvw.y - (int)uvw.y; u.z = uvw.z - (int)uvw.z;
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}
// v / 255.0f for a byte v, correctly rounded, without the division: one Newton step on v * RN(1/255).  Equal to the
// IEEE quotient for all 256 inputs (checked on the host when the library creates its first scene, u8_unorm_selftest).
__host__ __device__ __forceinline__ float u8_unorm(unsigned v) {
    const float x = (float)v, r = 1.0f / 255.0f;
    const float q0 = x * r;
    const float e = fmaf(-255.0f, q0, x);
    return fmaf(e, r, q0);
}
// Texel pointers come out of tables in memory (DTexture::data, DAtlas::data), so the compiler cannot know their address space
// and would fetch every texel with a flat load; they are global memory, and these loaders say so (PT_GLOBAL_AS, pt_device.h).
PT_DEV F3 texel(const DTexture& t, int y, int x) {
    const size_t i = (size_t)y * t.width + x;
    if (t.kind == MI355PT_TEX_RGB32F) { const pt_f4v v = ((const PT_GLOBAL_AS pt_f4v*)t.data)[i]; return f3(v.x, v.y, v.z); }
    const uint32_t v = ((const PT_GLOBAL_AS uint32_t*)t.data)[i];         // uchar4: r, g, b, -
    return f3(u8_unorm(v & 255u), u8_unorm((v >> 8) & 255u), u8_unorm((v >> 16) & 255u));
}
PT_DEV F3 tex_point(const DTexture& t, F3 uvw) {
    if (t.kind == MI355PT_TEX_CHECKER) {
        const F3 u = tile_clamp(uvw);
        if (u.x <= 0.5f) return u.y <= 0.5f ? ld3(t.c1) : ld3(t.c2);
        return u.y <= 0.5f ? ld3(t.c2) : ld3(t.c1);
    }
    const int width = t.width, height = t.height;
    if (width + height == 0) return f3(0, 0, 0);
    const F3 u = tile_clamp(uvw);
    const float x = width * u.x, y = height * u.y;
    int ix = (int)x, iy = (int)y;
    const float fx = x - ix, fy = y - iy;
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    int ixp = ix + 1; if (ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
    int iyp = iy + 1; if (iyp >= height) iyp -= height;
    return texel(t, iy, ix) * ((1 - fx) * (1 - fy)) + texel(t, iy, ixp) * (fx * (1 - fy)) +
           texel(t, iyp, ix) * ((1 - fx) * fy) + texel(t, iyp, ixp) * (fx * fy);
}
// (x, y) of the 32 elliptic filter taps of scene.h:169-181, the same for every scene; set once per device by mi355pt_scene_create
__constant__ float g_filter_taps[64];
PT_DEV F3 tex_filtered(const DScene& S, const DTexture& t, F3 uvw, F3 d0, F3 d1) {
    F3 c = tex_point(t, uvw);
    if (len2(d0) + len2(d1) == 0) return c;
    for (int i = 1; i < 32; i++) {
        // elliptic tap offsets r*sin, r*cos of scene.h:170-175 are pixel-independent constants:
        // tabulated on the host with libm (mi355pt.hip, init_filter_taps)
        const float x = g_filter_taps[2 * i], y = g_filter_taps[2 * i + 1];
        c = c + tex_point(t, uvw + x * d0 + y * d1);
    }
    return c / 32.0f;
}
PT_DEV F3 texmap_sample(const DScene& S, int map, F3 uvw, F3 d0, F3 d1) {
    const mi355pt_texmap& m = S.texmaps[map];
    if (m.texture < 0) return f3(0, 0, 0);
    const F3 pos = ld3(m.pos);
    const F3 u = m3mul(m.itm, uvw - pos);
    const F3 e0 = m3mul(m.itm, (d0 + uvw) - pos) - u;
    const F3 e1 = m3mul(m.itm, (d1 + uvw) - pos) - u;
    return tex_filtered(S, S.textures[m.texture], u, e0, e1);
}
// The four maps of a material from its atlas (DAtlas): tex_point / tex_filtered / texmap_sample above, with the coordinates,
// wrap and bilinear weights computed once per tap and eight channels accumulated instead of three.
struct AtlasTaps { F3 alb, nrm; float rough, metal; };
PT_DEV AtlasTaps atlas_point(const DAtlas& t, F3 uvw) {
    const int width = t.width, height = t.height;
    const F3 u = tile_clamp(uvw);
    const float x = width * u.x, y = height * u.y;
    int ix = (int)x, iy = (int)y;
    const float fx = x - ix, fy = y - iy;
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    int ixp = ix + 1; if (ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
    int iyp = iy + 1; if (iyp >= height) iyp -= height;
    const PT_GLOBAL_AS pt_u2v* td = (const PT_GLOBAL_AS pt_u2v*)t.data;
    const pt_u2v t00 = td[(size_t)iy * width + ix], t01 = td[(size_t)iy * width + ixp];
    const pt_u2v t10 = td[(size_t)iyp * width + ix], t11 = td[(size_t)iyp * width + ixp];
    const float w00 = (1 - fx) * (1 - fy), w01 = fx * (1 - fy), w10 = (1 - fx) * fy, w11 = fx * fy;
#define PT_AT(word, sh) (u8_unorm((t00.word >> sh) & 255u) * w00 + u8_unorm((t01.word >> sh) & 255u) * w01 + \
                         u8_unorm((t10.word >> sh) & 255u) * w10 + u8_unorm((t11.word >> sh) & 255u) * w11)
    AtlasTaps r;
    r.alb = f3(PT_AT(x, 0), PT_AT(x, 8), PT_AT(x, 16));
    r.nrm = f3(PT_AT(x, 24), PT_AT(y, 0), PT_AT(y, 8));
    r.rough = PT_AT(y, 16); r.metal = PT_AT(y, 24);
#undef PT_AT
    return r;
}
PT_DEV AtlasTaps atlas_sample(const DScene& S, const DAtlas& t, const mi355pt_texmap& m, F3 uvw, F3 d0, F3 d1) {
    const F3 pos = ld3(m.pos);
    const F3 u = m3mul(m.itm, uvw - pos);
    const F3 e0 = m3mul(m.itm, (d0 + uvw) - pos) - u;
    const F3 e1 = m3mul(m.itm, (d1 + uvw) - pos) - u;
    AtlasTaps c = atlas_point(t, u);
    if (len2(e0) + len2(e1) == 0) return c;
    for (int i = 1; i < 32; i++) {
        const float x = g_filter_taps[2 * i], y = g_filter_taps[2 * i + 1];
        const AtlasTaps p = atlas_point(t, u + x * e0 + y * e1);
        c.alb = c.alb + p.alb; c.nrm = c.nrm + p.nrm; c.rough = c.rough + p.rough; c.metal = c.metal + p.metal;
    }
    c.alb = c.alb / 32.0f; c.nrm = c.nrm / 32.0f; c.rough = c.rough / 32.0f; c.metal = c.metal / 32.0f;
    return c;
}
PT_DEV F3 textured_color(const DScene& S, const float* color, int map, const Surf& h) {
    const F3 c = ld3(color);
    return map >= 0 ? c * texmap_sample(S, map, h.uvw, h.du0, h.du1) : c;
}
PT_DEV F3 textured_color_srgb(const DScene& S, const float* color, int map, const Surf& h) {
    const F3 c = ld3(color);
    if (map < 0) return c;
    F3 mc = texmap_sample(S, map, h.uvw, h.du0, h.du1);
    mc = f3(a_powf(mc.x, 2.2f), a_powf(mc.y, 2.2f), a_powf(mc.z, 2.2f));
    return c * mc;
}
PT_DEV F3 sample_environment(const DScene& S, F3 dir) {
    const F3 c = ld3(S.env_color);
    if (S.env_map < 0) return c;
    const F3 uvw = f3(0.5f - a_atan2f(dir.x, dir.y) / (2.0f * 3.14159274f), 0.5f + a_asinf(dir.z * 1.0f) / 3.14159274f, 0.0f);
    const mi355pt_texmap& m = S.texmaps[S.env_map];
    if (m.texture < 0) return c * f3(0, 0, 0);
    return c * tex_point(S.textures[m.texture], m3mul(m.itm, uvw - ld3(m.pos)));
}

// ------------------------------------------------------------------ Disney (disneyBrdf.h)
PT_DEV float dmin(float a, float b) { return a > b ? b : a; }
PT_DEV float dmax(float a, float b) { return a < b ? b : a; }
PT_DEV float dclamp(float t, float l, float r) { t = dmin(r, t); t = dmax(l, t); return t; }
PT_DEV float mixf(float a, float b, float r) { return a * (1.0f - r) + b * r; }
PT_DEV F3 mixv(F3 a, F3 b, float r) { return a * (1.0f - r) + b * r; }
// GTR1 with log(a * a) supplied: `a` depends on the material only, so its logarithm is taken once per vertex
// (disney_finish) instead of in each of the six pdf / eval calls of the vertex.
PT_DEV float gtr1(float NdotH, float a, float log_a2) {
    if (a >= 1.0f) return kInvPI;
    const float a2 = a * a;
    const float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
    return (a2 - 1.0f) / (kPI * log_a2 * t);
}
PT_DEV float gtr2(float NdotH, float a) {
    const float a2 = a * a;
    const float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
    return a2 / (kPI * t * t);
}
PT_DEV float smith_ggx(float NdotV, float a) {
    const float a2 = a * a, b = NdotV * NdotV;
    return 1.0f / (NdotV + sqrtf(a2 + b - a2 * b));
}
PT_DEV float schlick(float u) {
    // The reference's `1.0 - u` is a double subtraction rounded to float (disneyBrdf.h:69).  For a float u that is the float
    // subtraction: the exact difference fits a double whenever |u| >= 2^-29, and below that it lies strictly inside (1 - 2^-29, 1 + 2^-29),
    // where no midpoint of two floats is -- rounding once or twice gives the same float.
    const float m = dclamp(1.0f - u, 0.0f, 1.0f);
    const float m2 = m * m;
    return m2 * m2 * m;
}
struct Disney {
    F3 base;
    float metallic, specular, roughness, specularTint, sheenTint, sheen, clearcoatGloss, clearcoat, subsurface, csw;
    float a_eval, log_eval, a_pdf, log_pdf;      // clearcoat GTR1 widths of DisneyEval / DisneyPdf and log(a * a) of each
    // DisneyEval's terms that depend on the material and on wo only (disney_view): a vertex evaluates the BRDF three times (light
    // sample, MIS sample, continuation) for one wo.  Each evaluation sits behind its own `NdotL <= 0` exit, so the compiler cannot
    // share them between the calls; computed once here they are the same expressions, hence the same bits.
    F3 c_spec0, c_sheen;
    float ro, ro2, f_wo, g_wo, gr_wo;
};
PT_DEV void disney_finish(Disney& s) {      // Clamp + InitCSW (disneyBrdf.h:108-131)
    s.metallic = dclamp(s.metallic, 0.001f, 0.999f); s.specular = dclamp(s.specular, 0.001f, 0.999f);
    s.roughness = dclamp(s.roughness, 0.001f, 0.999f); s.specularTint = dclamp(s.specularTint, 0.001f, 0.999f);
    s.sheenTint = dclamp(s.sheenTint, 0.001f, 0.999f); s.sheen = dclamp(s.sheen, 0.001f, 0.999f);
    s.clearcoatGloss = dclamp(s.clearcoatGloss, 0.001f, 0.999f); s.clearcoat = dclamp(s.clearcoat, 0.001f, 0.999f);
    s.subsurface = dclamp(s.subsurface, 0.001f, 0.999f);
    const F3 cd = s.base;
    const float cd_lum = dot(cd, f3(0.3f, 0.6f, 0.1f));
    const F3 c_tint = cd_lum > 0.0f ? (cd / cd_lum) : f3(1.0f, 1.0f, 1.0f);
    const F3 c_spec0 = mixv((1.0f - s.specular * 0.3f) * mixv(f3(1, 1, 1), c_tint, s.specularTint), cd, s.metallic);
    const float cs_lum = dot(c_spec0, f3(0.3f, 0.6f, 0.1f));
    s.csw = cs_lum / (cs_lum + (1.0f - s.metallic) * cd_lum);
    // (a_eval, log_eval, a_pdf, log_pdf: functions of the material's clearcoatGloss alone, set by mat_prepare from DMaterial)
    // DisneyEval's own c_spec0 (0.3 * specular, where InitCSW has 1 - 0.3 * specular: disneyBrdf.h:127 vs :166) and c_sheen
    s.c_spec0 = mixv(s.specular * 0.3f * mixv(f3(1, 1, 1), c_tint, s.specularTint), cd, s.metallic);
    s.c_sheen = mixv(f3(1, 1, 1), c_tint, s.sheenTint);
    s.ro = dmax(0.001f, s.roughness);
    const float r2 = s.roughness * 0.5f + 0.5f;
    s.ro2 = r2 * r2;
}
PT_DEV float disney_pdf(const Disney& s, float NdotH, float NdotL, float HdotL) {
    const float d_pdf = NdotL * (1.0f / kPI);
    const float r_pdf = gtr2(NdotH, dmax(0.001f, dmax(0.001f, s.roughness))) * NdotH / (4.0f * HdotL);
    const float c_pdf = gtr1(NdotH, s.a_pdf, s.log_pdf) * NdotH / (4.0f * HdotL);
    return c_pdf * s.clearcoat + (1.0f - s.clearcoat) * (s.csw * r_pdf + (1.0f - s.csw) * d_pdf);
}
// the wo-dependent terms of DisneyEval (disneyBrdf.h:176-205) for the vertex's NdotV
PT_DEV void disney_view(Disney& s, float NdotV) {
    s.f_wo = schlick(NdotV);
    s.g_wo = smith_ggx(NdotV, s.ro2);
    s.gr_wo = smith_ggx(NdotV, 0.25f);
}
PT_DEV F3 disney_eval(const Disney& s, float NdotL, float NdotV, float NdotH, float HdotL) {
    if (NdotL <= 0.0f || NdotV <= 0.0f) return f3(0, 0, 0);
    const F3 cd = s.base;
    const float f_wo = s.f_wo, f_wi = schlick(NdotL);
    const float fd90 = 0.5f + 2.0f * HdotL * HdotL * s.roughness;
    const float fd = mixf(1.0f, fd90, f_wo) * mixf(1.0f, fd90, f_wi);
    const float fss90 = HdotL * HdotL * s.roughness;
    const float fss = mixf(1.0f, fss90, f_wo) * mixf(1.0f, fss90, f_wi);
    const float ss = 1.25f * (fss * (1.0f / (NdotV + NdotL) - 0.5f) + 0.5f);
    const float ds = gtr2(NdotH, s.ro);
    const float fh = schlick(HdotL);
    const F3 fs = mixv(s.c_spec0, f3(1, 1, 1), fh);
    float gs = s.g_wo;
    gs *= smith_ggx(NdotL, s.ro2);
    const F3 f_sheen = fh * s.sheen * s.c_sheen;
    const float dr = gtr1(NdotH, s.a_eval, s.log_eval);
    const float fr = mixf(0.04f, 1.0f, fh);
    const float gr = s.gr_wo * smith_ggx(NdotL, 0.25f);
    const F3 f = ((1.0f / kPI) * mixf(fd, ss, s.subsurface) * cd + f_sheen) * (1.0f - s.metallic) + gs * fs * ds + (0.25f * s.clearcoat) * gr * fr * dr;
    return f * NdotL;
}
PT_DEV F3 reflect(F3 I, F3 N) { return I - 2.0f * dot(N, I) * N; }
PT_DEV F3 disney_sample(const Disney& s, F3 V, F3 N, Rng& rng) {
    float r1 = rng.U(), r2 = rng.U();
    const F3 U = fabsf(N.z) < (1.0f - kFltEps) ? f3(0.0f, 0.0f, 1.0f) : f3(1.0f, 0.0f, 0.0f);
    const F3 T = normalized(cross(U, N));
    const F3 B = cross(N, T);
    // Lobe choice first (disneyBrdf.h:222-268), then ONE sin/cos for whichever lobe each lane took: the three
    // lobes use the same expression for the azimuth, and a wave whose lanes disagree would otherwise run the
    // transcendental code three times.
    int lobe; float phi, cosTheta = 0.0f, sinTheta = 0.0f;
    if (r1 < s.clearcoat) {
        lobe = 0;
        r1 /= s.clearcoat;
        const float a = mixf(0.1f, 0.001f, s.clearcoatGloss);
        cosTheta = sqrtf((1.0f - a_powf(a * a, 1.0f - r2)) / (1.0f - a * a));
        sinTheta = sqrtf(dmax(0.0f, 1.0f - (cosTheta * cosTheta)));
        phi = r1 * kTwoPI;
    } else {
        r1 -= s.clearcoat;
        r1 /= (1.0f - s.clearcoat);
        if (r2 < s.csw) {
            lobe = 1;
            r2 /= s.csw;
            const float a = dmax(0.001f, s.roughness);
            cosTheta = sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
            sinTheta = sqrtf(dmax(0.0f, 1.0f - (cosTheta * cosTheta)));
            phi = r1 * kTwoPI;
        } else {
            lobe = 2;
            r2 -= s.csw;
            r2 /= (1.0f - s.csw);
            phi = kTwoPI * r2;
        }
    }
    float sin_p, cos_p;
    a_sincosf(phi, sin_p, cos_p);
    if (lobe < 2) {
        F3 H = normalized(f3(cos_p * sinTheta, sin_p * sinTheta, cosTheta));
        H = H.x * T + H.y * B + H.z * N;
        if (dot(H, V) <= 0.0f) H = H * -1.0f;
        return reflect(neg(V), H);
    }
    F3 d;
    const float r = sqrtf(r1);
    d.x = r * cos_p; d.y = r * sin_p;
    d.z = sqrtf(dmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return T * d.x + B * d.y + N * d.z;
}

// ------------------------------------------------------------------ Cook-Torrance (brdf_cook_torrance.h, utils.cpp:12-19,110-151)
PT_DEV void branchless_onb(F3 n, F3& b1, F3& b2) {
    const float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    b1 = f3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    b2 = f3(b, sign + n.y * n.y * a, -n.y);
}
PT_DEV F3 ggx_sample(float roughness, float& probability, Rng& rng) {
    const float a = roughness * roughness;
    const float F = rng.U() * 0.99999f;
    const float theta = a_acosf(sqrtf((1.0f - F) / (F * (a * a - 1.0f) + 1.0f)));
    float cosTheta, sinTheta;
    a_sincosf(theta, sinTheta, cosTheta);
    if (cosTheta < 0.0f) cosTheta = 0.0f;
    const float beta = rng.U() * kTwoPI;
    float bottom = (a * a - 1.0f) * cosTheta * cosTheta + 1.0f;
    bottom = bottom * bottom;
    if (bottom <= 0.0f) bottom = 0.001f;
    probability = a * a * cosTheta * sinTheta * kInvPI / bottom;
    if (probability <= 0.0f) probability = 0.001f;
    float sin_b, cos_b;
    a_sincosf(beta, sin_b, cos_b);
    return f3(sinTheta * cos_b, sinTheta * sin_b, cosTheta);
}
PT_DEV F3 cook_torrance(F3 L, F3 V, F3 N, F3 albedo, float roughness, float metalness) {
    const F3 H = normalized(V + L);
    F3 F0 = f3(0.04f, 0.04f, 0.04f);
    F0 = f3(F0.x * (1.0f - metalness) + albedo.x * metalness, F0.y * (1.0f - metalness) + albedo.y * metalness,
            F0.z * (1.0f - metalness) + albedo.z * metalness);
    const float VDotH = cymax(dot(V, H), 0.0f);
    const double q = (double)(1.0f - VDotH);
    const float pw = (float)((q * q) * (q * q) * q);     // pow(float, 5.0) evaluated in double (:105)
    const F3 F = F0 + neg(F0 - 1.0f) * pw;
    const float a = roughness * roughness, a2 = a * a;
    const float NdotH = cymax(dot(N, H), 0.0f);
    float denom = cymax((NdotH * NdotH * (a2 - 1.0f) + 1.0f), 0.001f);
    denom = kPI * denom * denom;
    const float NDF = a2 / denom;
    const float NdotV = cymax(dot(N, V), 0.0f), NdotL = cymax(dot(N, L), 0.0f);
    const float k = roughness * roughness * 0.5f;
    const float G = (NdotV / (NdotV * (1.0f - k) + k)) * (NdotL / (NdotL * (1.0f - k) + k));
    const F3 numerator = NDF * G * F;
    const float denominator = 4.0f * cymax(dot(N, V), 0.0f) * cymax(dot(N, L), 0.0f);
    const F3 specular = numerator / cymax(denominator, 0.001f);
    F3 kD = f3(1.0f, 1.0f, 1.0f) - F;
    kD = kD * (1.0f - metalness);
    return kD * albedo * kInvPI + specular;
}

// ------------------------------------------------------------------ material context
// The reference re-evaluates textures and the shading frame inside Sample, ComputePdf and every
// EvalBrdf call of a vertex (disneyMaterial.h:97-121 etc.); the values are deterministic, so they
// are evaluated once per vertex here.
struct MatCtx {
    int kind;
    Disney d;
    F3 brdfN, Ngeo, albedo, emission;
    float rough_raw, rough_eval, metal;
    float NdotV;                 // Disney: dot(brdfN, wo) of this vertex (mat_view)
};
// KIND >= 0: the caller knows the material kind (k_shade<KIND> only takes hits of that kind): the other model's code, and
// the registers its values would occupy, drop out at compile time.
// ATLAS: the scene has material atlases (DScene::mtl_atlas); instances for scenes without one do not carry that code.
template <int KIND = -1, bool ATLAS = true>
PT_DEV void mat_prepare(const DScene& S, const Surf& h, MatCtx& m) {
    const DMaterial& mt = S.materials[h.mtl];
    m.kind = KIND >= 0 ? KIND : mt.kind;
    F3 N = h.N; N = N / len(N);
    m.Ngeo = N;
    m.brdfN = N;
    const int atlas = (ATLAS && S.mtl_atlas) ? S.mtl_atlas[h.mtl] : -1;
    F3 base; float rough, metal;                         // the three textured colours, the same lookups for both models
    if (ATLAS && atlas >= 0) {                           // all four maps from one fused texture (DAtlas, pt_scene.h)
        const AtlasTaps a = atlas_sample(S, S.atlases[atlas], S.texmaps[mt.albedo_map], h.uvw, h.du0, h.du1);
        const F3 tn = f3(2.0f * a.nrm.x - 1.0f, 2.0f * a.nrm.y - 1.0f, 2.0f * a.nrm.z - 1.0f);
        F3 b = h.N * tn.z + normalized(h.B) * tn.y + normalized(h.T) * tn.x;
        m.brdfN = b / len(b);
        base = ld3(mt.albedo) * f3(a_powf(a.alb.x, 2.2f), a_powf(a.alb.y, 2.2f), a_powf(a.alb.z, 2.2f));
        rough = mt.roughness[0] * a.rough; metal = mt.metalness[0] * a.metal;
    } else {
        if (mt.normal_map >= 0) {
            const F3 c = texmap_sample(S, mt.normal_map, h.uvw, h.du0, h.du1);
            const F3 tn = f3(2.0f * c.x - 1.0f, 2.0f * c.y - 1.0f, 2.0f * c.z - 1.0f);
            F3 b = h.N * tn.z + normalized(h.B) * tn.y + normalized(h.T) * tn.x;
            m.brdfN = b / len(b);
        }
        base = textured_color_srgb(S, mt.albedo, mt.albedo_map, h);
        rough = textured_color(S, mt.roughness, mt.roughness_map, h).x;
        metal = textured_color(S, mt.metalness, mt.metalness_map, h).x;
    }
    m.emission = f3(0, 0, 0);
    if (m.kind == MI355PT_MTL_DISNEY) {
        m.d.base = base;
        m.d.roughness = rough;
        m.d.metallic = metal;
        m.d.clearcoat = mt.clearcoat; m.d.clearcoatGloss = mt.clearcoat_gloss; m.d.sheen = mt.sheen; m.d.sheenTint = mt.sheen_tint;
        m.d.specular = mt.specular; m.d.specularTint = mt.specular_tint; m.d.subsurface = mt.subsurface;
        m.d.a_eval = mt.a_eval; m.d.log_eval = mt.log_eval; m.d.a_pdf = mt.a_pdf; m.d.log_pdf = mt.log_pdf;
        disney_finish(m.d);
        const int li = h.light;
        if (li >= 0) m.emission = ld3(S.lights[li].intensity);      // disneyMaterial.h:193-198
    } else {
        m.albedo = base;
        m.rough_raw = rough;
        m.rough_eval = m.rough_raw <= 0.0f ? 0.001f : m.rough_raw;
        m.metal = metal;
    }
}
PT_DEV float mat_pdf(const MatCtx& m, F3 wi, F3 wo) {
    if (m.kind == MI355PT_MTL_DISNEY) {
        const F3 H = normalized(wi + wo);
        return disney_pdf(m.d, dot(m.brdfN, H), dot(m.brdfN, wi), dot(H, wi));
    }
    const float cosTheta = dot(wi, m.brdfN);
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    const float a = m.rough_raw * m.rough_raw;
    float bottom = (a * a - 1.0f) * cosTheta * cosTheta + 1.0f;
    bottom = bottom * bottom;
    return a * a * cosTheta * sinTheta * kInvPI / bottom;
}
PT_DEV void mat_sample(const MatCtx& m, F3 wo, Rng& rng, F3& wi, float& pdf) {
    if (m.kind == MI355PT_MTL_DISNEY) {
        wi = disney_sample(m.d, wo, m.brdfN, rng);
        pdf = mat_pdf(m, wi, wo);
    } else {
        F3 b1, b2;
        branchless_onb(m.brdfN, b1, b2);
        F3 sd = ggx_sample(m.rough_raw, pdf, rng);
        sd = m.brdfN * sd.z + b1 * sd.x + b2 * sd.y;
        wi = sd / len(sd);
    }
}
// once per vertex, after mat_prepare: the terms of the BRDF evaluations that depend on wo only
PT_DEV void mat_view(MatCtx& m, F3 wo) {
    if (m.kind == MI355PT_MTL_DISNEY) { m.NdotV = dot(m.brdfN, wo); disney_view(m.d, m.NdotV); }
}
PT_DEV F3 mat_eval(const MatCtx& m, F3 wi, F3 wo) {      // shadingNormal output == m.brdfN for both materials
    if (m.kind == MI355PT_MTL_DISNEY) {
        const F3 H = normalized(wi + wo);
        return m.emission + disney_eval(m.d, dot(m.brdfN, wi), m.NdotV, dot(m.brdfN, H), dot(H, wi));
    }
    return cook_torrance(wi, wo, m.Ngeo, m.albedo, m.rough_eval, m.metal);
}

// ------------------------------------------------------------------ area light (lightcomponent.cpp, objects.cpp:474-503)
// Object::Sample of the light's object (objbase.cpp:5-15, objects.cpp:474-486, model.h:69-76 + mesh.h:39-61)
PT_DEV void light_sample_point(const DScene& S, const DLight& L, Rng& rng, F3& sp, F3& sn) {
    const DNodeXf& n = S.nodes[L.node];
    F3 p, nn;
    if (S.plane_lights_only || L.kind == MI355PT_OBJ_PLANE) {
        const float x = rng.U() * 2.0f - 1.0f;
        const float y = rng.U() * 2.0f - 1.0f;
        p = f3(x, y, 0.0f); nn = f3(0.0f, 0.0f, 1.0f);
    } else if (L.kind == MI355PT_OBJ_MODEL) {
        const float* cd = S.light_data + L.data;
        int meshId = 0;
        if (L.n_cd != 1) {                                     // CDF::Sample, utils.cpp:227-246
            const float random = rng.U() * cd[L.n_cd - 1];
            for (int i = 1; i < L.n_cd; i++)
                if (random <= cd[i] && random >= cd[i - 1]) { meshId = i - 1; break; }
        }
        const float u0 = rng.U(), u1 = rng.U();                // UniformSampleTriangle, utils.cpp:201-207
        const float su0 = sqrtf(u0);
        const float b0 = 1.0f - su0, b1 = u1 * su0;
        const float* v = cd + L.n_cd + 9 * meshId;             // face 0 of the mesh: the stored mesh's own cdf is empty (model.h:51)
        const F3 p0 = ld3(v), p1 = ld3(v + 3), p2 = ld3(v + 6);
        p = b0 * p0 + b1 * p1 + (1.0f - b0 - b1) * p2;
        nn = gnormalize(gcross(p1 - p0, p2 - p0));
    } else {                                                   // Object::Sample: the origin, not transformed
        sp = f3(0.0f, 0.0f, 0.0f); sn = f3(0.0f, 0.0f, 0.0f);
        return;
    }
    sp = m4mulp(n.l2w, p, 1.0f);
    sn = normalized(m4mulp(n.l2w, nn, 1.0f) - m4mulp(n.l2w, f3(0.0f, 0.0f, 0.0f), 1.0f));
}
PT_DEV float light_pdf(const DLight& L, F3 hitp, F3 sp, F3 sn, float distance) {
    const float d2 = distance * distance;
    const F3 nwi = normalized(hitp - sp);
    const float c = cymax(dot(nwi, sn), 0.0001f);
    return d2 / (L.area * c);
}

// First sweep of LightVisTest (raytracer.cpp:67-97): the light's own object, both sides.  Returns true for a
// front-side hit; lp / ln = the hit point / normal in world space.  (The second sweep, over every other object,
// is k_trace<1> in mi355pt.hip.)  A Model's closest hit was found by k_trace<2> and is handed in as `mh`.
PT_DEV bool light_hit_point(const DScene& S, const Ray& wr, const DLight& L, const HitRec& mh, F3& lp, F3& ln, float* lz_out = nullptr) {
    const DInst& li = S.insts[L.inst];
    lp = ln = f3(0, 0, 0);
    Surf h;
    float lz = 0.0f; bool front; F3 P;
    if (S.plane_lights_only || L.kind == MI355PT_OBJ_PLANE) {
        const Ray lr = to_inst(S, li, wr);
        if (!plane_hit(lr.p, lr.d, SIDE_BOTH, lz, front, P)) return false;
        h.N = front ? f3(0.0f, 0.0f, 1.0f) : -1.0f * f3(0.0f, 0.0f, 1.0f);
    } else if (L.kind == MI355PT_OBJ_MODEL) {
        if (mh.inst < 0) return false;
        surf_from_hit(S, mh, wr, false, f3(0, 0, 0), f3(0, 0, 0), h);
        lp = h.p; ln = h.N;
        return h.front;
    } else {
        const Ray lr = to_inst(S, li, wr);
        if (!sphere_hit(lr.p, lr.d, SIDE_BOTH, lz, front, P)) return false;
        h.N = front ? P : -1.0f * P;                           // objects.cpp:685-790
    }
    h.p = P;
    if (lz_out) *lz_out = lz;
    surf_to_world(S, li, h, false);
    lp = h.p; ln = h.N;
    return front;
}
// Front / back of a Model light's recorded closest hit (IntersectRayWithFace, model.h:262-266), for k_trace<1>'s set-up
PT_DEV bool model_hit_front(const DScene& S, const DLight& L, const HitRec& mh, const Ray& wr) {
    const Ray lr = to_inst(S, S.insts[L.inst], wr);
    const Tri T = load_tri(S.tris, mh.slot);
    return gdot(lr.d, gcross(T.v1 - T.v0, T.v2 - T.v0)) < 0.0f;
}

// ------------------------------------------------------------------ tone mapping (tonemapping.h:14-42)
PT_DEV F3 aces(F3 v) {
    const float mi[9] = {0.59719f, 0.07600f, 0.02840f, 0.35458f, 0.90834f, 0.13383f, 0.04823f, 0.01566f, 0.83777f};
    const float mo[9] = {1.60475f, -0.10208f, -0.00327f, -0.53108f, 1.10813f, -0.07276f, -0.07367f, -0.00605f, 1.07602f};
    v = m3mul(mi, v);
    const F3 a = v * (v + 0.0245786f) - 0.000090537f;
    const F3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    v = a / b;
    v = m3mul(mo, v);
    v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y; v.z = v.z < 0.0f ? 0.0f : v.z;
    v.x = v.x > 1.0f ? 1.0f : v.x; v.y = v.y > 1.0f ? 1.0f : v.y; v.z = v.z > 1.0f ? 1.0f : v.z;
    return v;
}

}  // namespace ptd
