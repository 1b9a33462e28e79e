// cloth_render.hpp -- headless RGB / depth rasteriser of the cloth mesh.
//
// The reference renders its image observations by exporting the particle grid as a triangle mesh
// (cloth_env.py:218-229: faces (pp, pp+wh, pp+1) and (pp+1, pp+wh, pp+wh+1) per grid quad) and calling a Blender
// subprocess (gym_cloth/blender/get_image_rep_279.py: pinhole camera at (0.5, 0.5, 1.45) looking straight down, lens 40 mm on a
// 36 mm sensor, optional camera jitter; the two sides of the cloth in different colours, :241-257, swapped for a tier-2 cloth
// dropped from the other side, :235-239; a constant-falloff lamp without shadows, :448-450; a white bed plane under the
// cloth, :159-172; for depth images the Z pass normalised over the image, :390-406). This file rasterises the same scene on the
// GPU, one workgroup per cloth, straight from the SoA particle state -- no mesh export, no subprocess:
//   vertices  -> camera space -> pixel coordinates (perspective divide), per-vertex normals from the incident faces
//   triangles -> edge functions over their bounding boxes (pixel centres, top-left fill rule), perspective-correct depth,
//                smooth-shaded two-sided Lambert colour; nearest fragment wins through a 64-bit atomicMax on
//                (depth key << 32 | rgb)
//   resolve   -> uint8 RGB [H][W][3] and float camera-space depth [H][W] (background: the bed plane z = 0)
// It does not try to reproduce Blender's pixels (no renderer to compare with exists outside Blender); what it is pinned to is
// oracle/render_oracle.py, a numpy restatement of exactly these rules in float32, bit for bit (tests/test_gpu_render.py).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/clothhip.h"

namespace clothhip {

struct RenderArgs {
    int32_t N, P, Ppad, W, H, E;
    float cam[3];          // camera position
    float R[9];            // world -> camera rotation, row major (camera looks along -z_cam, +y_cam is up)
    float fx, fy, cx, cy;  // pixels: u = fx * x_cam / (-z_cam) + cx, v = cy - fy * y_cam / (-z_cam)
    float front[3], back[3], bg[3];
    float light[3];        // unit vector TOWARDS the lamp
    float ambient, energy;
    const uint8_t *swap;   // [E] or nullptr: != 0 swaps the two side colours (tier-2 cloth with init_side False)
    unsigned long long *zbuf;   // [E][H*W] scratch
    uint8_t *rgb;          // [E][H][W][3] or nullptr
    float *depth;          // [E][H][W] or nullptr
};

__device__ __forceinline__ uint32_t quant8(float v) {
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return (uint32_t)(v * 255.0f + 0.5f);
}

// key of a fragment: larger = nearer. d = camera-space depth (> 0): float bits of 1/d are monotone for d > 0.
__device__ __forceinline__ uint32_t depth_key(float inv_d) { return __float_as_uint(inv_d); }

template <typename T>
__global__ __launch_bounds__(256) void k_render(const T *pos, RenderArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *vx = reinterpret_cast<float *>(smem), *vy = vx + A.Ppad, *vd = vy + A.Ppad, *vi = vd + A.Ppad;   // pixel x, y, depth, intensity
    float *wx = vi + A.Ppad, *wy = wx + A.Ppad, *wz = wy + A.Ppad;                                          // world position
    const int e = blockIdx.x, tid = threadIdx.x, N = A.N, P = A.P;
    const T *px = pos + (size_t)e * 3 * A.Ppad, *py = px + A.Ppad, *pz = py + A.Ppad;
    unsigned long long *zb = A.zbuf + (size_t)e * A.W * A.H;
    const float bed_d = A.cam[2];                                    // camera-space depth of the bed plane z = 0 (top-down)
    // background: depth key 0 = behind everything (a cloth lying ON the bed plane must not z-fight with it); its reported
    // depth is the bed plane's
    const unsigned long long bgkey = (unsigned long long)((quant8(A.bg[0]) << 16) | (quant8(A.bg[1]) << 8) | quant8(A.bg[2]));
    for (int i = tid; i < A.W * A.H; i += 256) zb[i] = bgkey;
    for (int i = tid; i < P; i += 256) { wx[i] = (float)px[i]; wy[i] = (float)py[i]; wz[i] = (float)pz[i]; }
    __syncthreads();
    // ---- vertices: projection + smooth normal (sum of the incident faces' normals, in face index order) ----------------------
    for (int i = tid; i < P; i += 256) {
        const float X = wx[i] - A.cam[0], Y = wy[i] - A.cam[1], Z = wz[i] - A.cam[2];
        const float xc = A.R[0] * X + A.R[1] * Y + A.R[2] * Z;
        const float yc = A.R[3] * X + A.R[4] * Y + A.R[5] * Z;
        const float zc = A.R[6] * X + A.R[7] * Y + A.R[8] * Z;
        const float d = -zc;                                          // depth along the view axis
        const float ds = d > 1e-6f ? d : 1e-6f;
        vx[i] = (A.fx * xc) / ds + A.cx;
        vy[i] = A.cy - (A.fy * yc) / ds;
        vd[i] = d;
        const int r = i / N, c = i - r * N;
        float nx = 0.0f, ny = 0.0f, nz = 0.0f;
        // the quads (qr, qc) around the vertex, each with its two faces (cloth_env.py:224-229)
        for (int qr = r - 1; qr <= r; qr++)
            for (int qc = c - 1; qc <= c; qc++) {
                if (qr < 0 || qc < 0 || qr >= N - 1 || qc >= N - 1) continue;
                const int pp = qr * N + qc;
                const int f[2][3] = {{pp, pp + N, pp + 1}, {pp + 1, pp + N, pp + N + 1}};
                for (int k = 0; k < 2; k++) {
                    if (f[k][0] != i && f[k][1] != i && f[k][2] != i) continue;
                    const int a = f[k][0], b = f[k][1], cc = f[k][2];
                    const float ux = wx[b] - wx[a], uy = wy[b] - wy[a], uz = wz[b] - wz[a];
                    const float tx = wx[cc] - wx[a], ty = wy[cc] - wy[a], tz = wz[cc] - wz[a];
                    nx = nx + (uy * tz - uz * ty); ny = ny + (uz * tx - ux * tz); nz = nz + (ux * ty - uy * tx);
                }
            }
        const float nn = sqrtf(nx * nx + ny * ny + nz * nz);
        float lam = 0.0f;
        if (nn > 0.0f) {
            lam = (nx * A.light[0] + ny * A.light[1] + nz * A.light[2]) / nn;
            lam = lam < 0.0f ? -lam : lam;                            // two-sided
        }
        vi[i] = A.ambient + A.energy * lam;
    }
    __syncthreads();
    // ---- triangles --------------------------------------------------------------------------------------------------------
    const int nq = (N - 1) * (N - 1);
    const bool sw = A.swap != nullptr && A.swap[e] != 0;
    for (int t = tid; t < 2 * nq; t += 256) {
        const int q = t >> 1, qr = q / (N - 1), qc = q - qr * (N - 1), pp = qr * N + qc;
        const int a = (t & 1) ? pp + 1 : pp, b = pp + N, c = (t & 1) ? pp + N + 1 : pp + 1;
        const float x0 = vx[a], y0 = vy[a], x1 = vx[b], y1 = vy[b], x2 = vx[c], y2 = vy[c];
        if (!(vd[a] > 1e-6f && vd[b] > 1e-6f && vd[c] > 1e-6f)) continue;          // behind the camera: dropped
        const float area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
        if (!(area > 0.0f) && !(area < 0.0f)) continue;                            // degenerate (or NaN)
        // which side of the cloth faces the camera: the sign of the projected area (faces are wound alike; image y points
        // down, so the faces of the flat start grid, whose normals point up at the camera, project with negative area)
        const bool front = (area < 0.0f) != sw;
        const float *col = front ? A.front : A.back;
        float mnx = fminf(x0, fminf(x1, x2)), mxx = fmaxf(x0, fmaxf(x1, x2));
        float mny = fminf(y0, fminf(y1, y2)), mxy = fmaxf(y0, fmaxf(y1, y2));
        mnx = fmaxf(mnx, -1.0f); mny = fmaxf(mny, -1.0f); mxx = fminf(mxx, (float)A.W); mxy = fminf(mxy, (float)A.H);
        int ix0 = (int)floorf(mnx), ix1 = (int)floorf(mxx), iy0 = (int)floorf(mny), iy1 = (int)floorf(mxy);
        ix0 = ix0 < 0 ? 0 : ix0; iy0 = iy0 < 0 ? 0 : iy0;
        ix1 = ix1 > A.W - 1 ? A.W - 1 : ix1; iy1 = iy1 > A.H - 1 ? A.H - 1 : iy1;
        const float s = area > 0.0f ? 1.0f : -1.0f;                   // orient the edge functions so that inside is >= 0
        const float iw0 = 1.0f / vd[a], iw1 = 1.0f / vd[b], iw2 = 1.0f / vd[c];
        for (int iy = iy0; iy <= iy1; iy++)
            for (int ix = ix0; ix <= ix1; ix++) {
                const float fxp = (float)ix + 0.5f, fyp = (float)iy + 0.5f;
                const float e0 = s * ((x2 - x1) * (fyp - y1) - (y2 - y1) * (fxp - x1));   // opposite vertex a
                const float e1 = s * ((x0 - x2) * (fyp - y2) - (y0 - y2) * (fxp - x2));   // opposite vertex b
                const float e2 = s * ((x1 - x0) * (fyp - y0) - (y1 - y0) * (fxp - x0));   // opposite vertex c
                // top-left rule on exact zeros: an edge owns its pixels if it is a top or a left edge
                const bool in0 = e0 > 0.0f || (e0 == 0.0f && ((s * (y2 - y1) > 0.0f) || (y2 == y1 && s * (x2 - x1) < 0.0f)));
                const bool in1 = e1 > 0.0f || (e1 == 0.0f && ((s * (y0 - y2) > 0.0f) || (y0 == y2 && s * (x0 - x2) < 0.0f)));
                const bool in2 = e2 > 0.0f || (e2 == 0.0f && ((s * (y1 - y0) > 0.0f) || (y1 == y0 && s * (x1 - x0) < 0.0f)));
                if (!(in0 && in1 && in2)) continue;
                const float sa = s * area;
                const float b0 = e0 / sa, b1 = e1 / sa, b2 = e2 / sa;
                const float inv_d = b0 * iw0 + b1 * iw1 + b2 * iw2;                        // perspective-correct 1/depth
                const float inten = b0 * vi[a] + b1 * vi[b] + b2 * vi[c];
                const unsigned long long key = ((unsigned long long)depth_key(inv_d) << 32) | (quant8(col[0] * inten) << 16) |
                                               (quant8(col[1] * inten) << 8) | quant8(col[2] * inten);
                atomicMax(&zb[iy * A.W + ix], key);
            }
    }
    __syncthreads();
    // ---- resolve ----------------------------------------------------------------------------------------------------------
    for (int i = tid; i < A.W * A.H; i += 256) {
        const unsigned long long k = atomicMax(&zb[i], 0ull);       // read through L2, where the fragment atomics landed
        if (A.rgb) {
            uint8_t *o = A.rgb + ((size_t)e * A.W * A.H + i) * 3;
            o[0] = (uint8_t)((k >> 16) & 0xFF); o[1] = (uint8_t)((k >> 8) & 0xFF); o[2] = (uint8_t)(k & 0xFF);
        }
        if (A.depth) A.depth[(size_t)e * A.W * A.H + i] = (k >> 32) == 0ull ? bed_d : 1.0f / __uint_as_float((uint32_t)(k >> 32));
    }
}

}  // namespace clothhip
