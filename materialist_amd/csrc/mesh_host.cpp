// mesh_host.cpp -- the reference's depth -> mesh conversion (scene preparation), HOST code of libmatpbr.so.
//
// Restates `depth_file_to_mesh` -> `detect_boundary_points` (myutils/mesh_recon.py:41-74,86-331) and the 180-degree rotation about x of
// inverse_img_w_mi.py:726-727.  The reference walks every pixel in Python (minutes at 512 x 512); the three passes below are the same
// sequential algorithm in C++ (milliseconds), so the gap closing at depth discontinuities -- which depends on the scan order: a pixel
// gets ONE duplicate vertex, placed by the first triangle that asks for it -- comes out vertex for vertex and triangle for triangle
// (tests/golden/mesh_normals.npz holds the reference's own output).
//   pass 1 (:107-156)  a pixel whose fan of four triangles contains one seen at less than `min_angle` from its viewing ray, and which
//                      is nearer than the neighbours spanning that triangle, is a foreground boundary pixel; it REFERS to the deepest
//                      such neighbour;
//   pass 2 (:158-175)  every pixel follows its chain of references to the end and takes that pixel's depth (foreground silhouettes are
//                      pushed back onto the background, which removes the sliver triangles across the edge);
//   pass 3 (:177-300)  two triangles per cell; a triangle still seen at a grazing angle gets its nearest vertex (then its second
//                      nearest) replaced by a duplicate of that pixel at the triangle's largest depth.
// The per-pixel geometric normal the kernels shade with is the area-weighted normal of the grid vertex over its
// triangles, oriented towards the camera; a grid vertex without triangles has no geometry (its camera ray sees the environment).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/matpbr.h"

namespace {

struct V3 { double x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline bool is_zero(V3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }

// angle (degrees) between a triangle and the ray to its centre; NaN for a degenerate triangle (every comparison with it is false,
// as in the reference's numpy arithmetic)
inline double view_angle(V3 a, V3 b, V3 c) {
    V3 n = cross(sub(a, b), sub(a, c));
    const double ln = std::sqrt(dot(n, n));
    n = {n.x / ln, n.y / ln, n.z / ln};
    V3 ctr = {(a.x + b.x + c.x) / 3.0, (a.y + b.y + c.y) / 3.0, (a.z + b.z + c.z) / 3.0};
    const double lc = std::sqrt(dot(ctr, ctr));
    const double s = std::fabs(dot(n, {ctr.x / lc, ctr.y / lc, ctr.z / lc}));
    return std::asin(s) * (180.0 / M_PI);
}

}  // namespace

extern "C" int matpbr_depth_to_mesh_host(const float* depth_in, int H, int W, float fov_x_deg, float min_angle_deg, float* new_depth, double* vertices,
                                         int* n_vertices, int* triangles, int* n_triangles, float* normals) {
    if (!depth_in || !new_depth || !vertices || !n_vertices || !triangles || !n_triangles || H < 2 || W < 2) return MATPBR_ERR_INVALID_ARG;
    if (!(fov_x_deg > 0.0f && fov_x_deg < 179.0f)) return MATPBR_ERR_INVALID_ARG;
    const double f = (0.5 * W) / std::tan(0.5 * (double)fov_x_deg * M_PI / 180.0), cx = 0.5 * (W - 1), cy = 0.5 * (H - 1);
    const double min_angle = min_angle_deg;
    const long P = (long)H * W;
    std::vector<float> d(depth_in, depth_in + P);
    auto D = [&](int i, int j) -> float& { return d[(long)i * W + j]; };
    auto cam = [&](int i, int j, double dd) -> V3 { return {(j - cx) / f * dd, (i - cy) / f * dd, dd}; };   // K^-1 [j, i, 1] depth  (:101,107)
    // ---- pass 1: references of the foreground boundary pixels
    std::vector<int> ref_i(P, -1), ref_j(P, -1);
    const int di[4] = {1, -1, -1, 1}, dj[4] = {1, 1, -1, -1};                       // `direction` (:123)
    for (int i = 1; i < H - 1; ++i)
        for (int j = 1; j < W - 1; ++j) {
            const V3 v[5] = {cam(i, j, D(i, j)), cam(i + 1, j, D(i + 1, j)), cam(i, j + 1, D(i, j + 1)), cam(i - 1, j, D(i - 1, j)), cam(i, j - 1, D(i, j - 1))};
            if (is_zero(v[0]) || is_zero(v[1]) || is_zero(v[2]) || is_zero(v[3]) || is_zero(v[4])) continue;
            const int comb[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 4}, {0, 4, 1}};
            int ri = -1, rj = -1;
            for (int q = 0; q < 4; ++q) {
                const double ang = view_angle(v[comb[q][0]], v[comb[q][1]], v[comb[q][2]]);
                if (!(ang < min_angle)) continue;
                const float dh = D(i, j + dj[q]), dv = D(i + di[q], j);
                if (!(D(i, j) < dh || D(i, j) < dv)) continue;
                int li, lj;
                if (dh > dv) { li = i; lj = j + dj[q]; } else { li = i + di[q]; lj = j; }
                if (ri >= 0 && rj >= 0) {
                    if (D(ri, rj) < D(li, lj)) { ri = li; rj = lj; }
                } else { ri = li; rj = lj; }
            }
            ref_i[(long)i * W + j] = ri;
            ref_j[(long)i * W + j] = rj;
        }
    // ---- pass 2: every pixel takes the depth at the end of its chain of references (in place, row by row, as the reference does; the
    // ends of the chains are never modified themselves)
    for (int i = 1; i < H - 1; ++i)
        for (int j = 1; j < W - 1; ++j) {
            int ci = i, cj = j;
            while (ref_i[(long)ci * W + cj] != -1 && ref_j[(long)ci * W + cj] != -1) {
                const int ti = ci;
                ci = ref_i[(long)ci * W + cj];
                cj = ref_j[(long)ti * W + cj];
            }
            D(i, j) = D(ci, cj);
        }
    std::memcpy(new_depth, d.data(), sizeof(float) * (size_t)P);
    // ---- pass 3: triangles, duplicates where a triangle is still seen at a grazing angle
    std::vector<V3> copies;
    std::vector<int> copy_of(P, -1);
    int nt = 0;
    auto emit = [&](const long* id) {
        triangles[3 * (long)nt] = (int)id[0]; triangles[3 * (long)nt + 1] = (int)id[1]; triangles[3 * (long)nt + 2] = (int)id[2];
        ++nt;
    };
    auto valid = [&](const V3* v) { return view_angle(v[0], v[1], v[2]) > min_angle; };
    // one triangle: corner pixels (yi, xi); false when a corner has no depth (the caller then leaves the cell, as the reference's `continue` does)
    auto triangle = [&](const int* yi, const int* xi) -> bool {
        V3 v[3];
        float vd[3];
        long id[3];
        for (int k = 0; k < 3; ++k) {
            vd[k] = D(yi[k], xi[k]);
            v[k] = cam(yi[k], xi[k], vd[k]);
            id[k] = (long)yi[k] * W + xi[k];
            if (is_zero(v[k])) return false;
        }
        if (valid(v)) { emit(id); return true; }
        float largest = vd[0] > vd[1] ? vd[0] : vd[1];
        largest = largest > vd[2] ? largest : vd[2];
        for (int attempt = 0; attempt < 2; ++attempt) {
            int c = 0;                                        // np.argmin: the first of the smallest
            if (vd[1] < vd[c]) c = 1;
            if (vd[2] < vd[c]) c = 2;
            const long pix = (long)yi[c] * W + xi[c];
            if (copy_of[pix] < 0) {                           // one duplicate per pixel: the first request places it (:208-216)
                copy_of[pix] = (int)copies.size();
                copies.push_back(cam(yi[c], xi[c], largest));
            }
            v[c] = copies[copy_of[pix]];
            if (attempt == 0) vd[c] = largest;
            id[c] = P + copy_of[pix];
            if (valid(v)) { emit(id); return true; }
        }
        return true;
    };
    for (int i = 0; i < H - 1; ++i)
        for (int j = 0; j < W - 1; ++j) {
            const int y1[3] = {i, i + 1, i}, x1[3] = {j, j, j + 1};
            if (!triangle(y1, x1)) continue;                  // a corner without depth: the reference skips the rest of the cell too (:187-188)
            const int y2[3] = {i, i + 1, i + 1}, x2[3] = {j + 1, j, j + 1};
            triangle(y2, x2);
        }
    // ---- vertices in the renderer's frame: rotated 180 degrees about x (inverse_img_w_mi.py:726-727), grid first, duplicates after
    const long NV = P + (long)copies.size();
    for (long p = 0; p < P; ++p) {
        const V3 c = cam((int)(p / W), (int)(p % W), d[p]);
        vertices[3 * p] = c.x; vertices[3 * p + 1] = -c.y; vertices[3 * p + 2] = -c.z;
    }
    for (size_t k = 0; k < copies.size(); ++k) {
        vertices[3 * (P + (long)k)] = copies[k].x; vertices[3 * (P + (long)k) + 1] = -copies[k].y; vertices[3 * (P + (long)k) + 2] = -copies[k].z;
    }
    *n_vertices = (int)NV;
    *n_triangles = nt;
    if (normals) {     // area-weighted vertex normals of the grid vertices, towards the camera; zero where a vertex has no triangle
        std::vector<double> acc(3 * (size_t)P, 0.0);
        for (long t = 0; t < nt; ++t) {
            const int* id = triangles + 3 * t;
            const V3 a = {vertices[3 * (long)id[0]], vertices[3 * (long)id[0] + 1], vertices[3 * (long)id[0] + 2]};
            const V3 b = {vertices[3 * (long)id[1]], vertices[3 * (long)id[1] + 1], vertices[3 * (long)id[1] + 2]};
            const V3 c = {vertices[3 * (long)id[2]], vertices[3 * (long)id[2] + 1], vertices[3 * (long)id[2] + 2]};
            const V3 fn = cross(sub(b, a), sub(c, a));
            for (int k = 0; k < 3; ++k)
                if (id[k] < P) { acc[3 * (long)id[k]] += fn.x; acc[3 * (long)id[k] + 1] += fn.y; acc[3 * (long)id[k] + 2] += fn.z; }
        }
        for (long p = 0; p < P; ++p) {
            const double ax = acc[3 * p], ay = acc[3 * p + 1], az = acc[3 * p + 2];
            const double ln = std::sqrt(ax * ax + ay * ay + az * az);
            const double sgn = (ax * vertices[3 * p] + ay * vertices[3 * p + 1] + az * vertices[3 * p + 2]) > 0.0 ? -1.0 : 1.0;
            normals[3 * p] = ln > 0.0 ? (float)(sgn * ax / ln) : 0.0f;
            normals[3 * p + 1] = ln > 0.0 ? (float)(sgn * ay / ln) : 0.0f;
            normals[3 * p + 2] = ln > 0.0 ? (float)(sgn * az / ln) : 0.0f;
        }
    }
    return MATPBR_OK;
}
