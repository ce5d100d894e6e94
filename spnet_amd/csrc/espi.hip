// Synthetic "fake-ESPI" frames rasterised directly in HBM (the reference generator is
// gen_fake_espi.py:60-279: draw_waves :60-80, draw_rings :101-114, draw_antinodes :145-206, gen_images :217-279,
// up to but excluding the band-pass mix-up with the author's private images).
//
// The host draws the PARAMETERS of a frame (wave train, antinode list incl. the non-overlap rejection loop: a few
// dozen random numbers) with the same RandomState recipe as spnet_amd/fake_espi.py; this kernel turns them into
// pixels analytically -- one thread per pixel, no raster library, no PNG round trip -- and applies the sensor model
// (additive N(40,40) noise clipped to uint8, 50 % pixel dropout) from a counter-based RNG:
//   canvas 128 -> black wavy bands (polylines of `thick` px through y_j(x) = j*spacing - W|slope| + slope*x
//   + amp*cos(x/wavelength)) -> per antinode 2*rings concentric ellipse outlines, alternately black / 138, of
//   thickness round(min(a,b)/(2*rings)), later outlines over earlier ones -> + noise, saturate -> * {0,1}.
// Frames are statistically, not bitwise, those of the OpenCV rasteriser (outline distance is the first-order
// distance |F-1|/|grad F| to the implicit ellipse / the band's normal distance |dy|/sqrt(1+y'^2)).
#include "common.h"

#define ESPI_MAX_NODES 7
#define ESPI_NODE_STRIDE 8    // cx, cy, a, b, angle_deg, rings, start (0/1), valid
#define ESPI_WAVE_STRIDE 5    // amp, wavelength, thickness, slope, spacing

__device__ __forceinline__ unsigned espi_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void fake_espi_kernel(const float* __restrict__ waves,
                                                        const float* __restrict__ nodes,
                                                        const int* __restrict__ nnode, int H, int W,
                                                        unsigned seed, int noise, float* __restrict__ out_f,
                                                        unsigned char* __restrict__ out_u8) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
  if (x >= W) return;
  float val = 128.f;
  {   // wave bands
    const float* wv = waves + (long)f * ESPI_WAVE_STRIDE;
    const float amp = wv[0], wl = wv[1], thick = wv[2], slope = wv[3], spacing = wv[4];
    const float xf = (float)x;
    const float base = slope * xf + amp * cosf(xf / wl) - (float)W * fabsf(slope);
    const float dydx = slope - amp / wl * sinf(xf / wl);
    const float j = rintf(((float)y - base) / spacing);
    const int nl = 60 + (int)((float)H / spacing);
    if (j >= 0.f && j < (float)nl) {
      const float dist = fabsf((float)y - (base + j * spacing)) * rsqrtf(1.f + dydx * dydx);
      if (dist <= 0.5f * thick) val = 0.f;
    }
  }
  const int nn = nnode[f];
  for (int a = 0; a < nn; ++a) {   // antinodes in drawing order; inside one, the outermost matching outline wins
    const float* nd = nodes + ((long)f * ESPI_MAX_NODES + a) * ESPI_NODE_STRIDE;
    if (nd[7] == 0.f) continue;
    const float cx = nd[0], cy = nd[1], A = nd[2], Bx = nd[3];
    const float th = -nd[4] * 0.017453292519943295f;     // the reference passes -angle to cv2.ellipse (utils.py:50)
    const float dx = (float)x - cx, dy = (float)y - cy;
    if (dx * dx + dy * dy > (A + 8.f) * (A + 8.f)) continue;
    const float cs = cosf(th), sn = sinf(th);
    const float u = dx * cs + dy * sn, v = -dx * sn + dy * cs;
    const int rings = (int)nd[5];
    const int nwb = max(2 * rings, 1);
    const float t = fmaxf(rintf(fminf(A, Bx) / (float)nwb), 1.f);
    const int start = (int)nd[6];
    const float rho = sqrtf((u / A) * (u / A) + (v / Bx) * (v / Bx));
    // outlines j (0-based) sit at rho_j = (j+1)/(nwb+1); test the few around rho, highest j first
    const int jc = (int)floorf(rho * (float)(nwb + 1)) - 1;
    for (int j = min(jc + 2, nwb - 1); j >= max(jc - 1, 0); --j) {
      const float s = (float)(j + 1) / (float)(nwb + 1);
      const float aj = A * s, bj = Bx * s;
      const float gu = u / (aj * aj), gv = v / (bj * bj);
      const float F = sqrtf(u * gu + v * gv);            // sqrt((u/aj)^2 + (v/bj)^2)
      const float gn = sqrtf(gu * gu + gv * gv);
      const float dist = gn > 0.f ? fabsf(F - 1.f) * F / gn : bj;
      if (dist <= 0.5f * t) {
        val = ((start + j) & 1) ? 138.f : 0.f;
        break;
      }
    }
  }
  if (noise) {
    const unsigned pix = ((unsigned)f * (unsigned)H + (unsigned)y) * (unsigned)W + (unsigned)x;
    const unsigned h1 = espi_hash(pix * 0x9e3779b9u + seed), h2 = espi_hash(h1 ^ 0x85ebca6bu), h3 = espi_hash(h2 + 0xc2b2ae35u);
    const float u1 = ((float)(h1 >> 8) + 1.f) * (1.f / 16777217.f), u2 = (float)(h2 >> 8) * (1.f / 16777216.f);
    const float n = 40.f + 40.f * sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
    val = fminf(val + fminf(fmaxf(rintf(n), 0.f), 255.f), 255.f);      // cv2.randn into uint8 saturates; cv2.add too
    if (h3 & 0x10000u) val = 0.f;                                      // np.random.choice([0,1]) mask
  }
  const long o = ((long)f * H + y) * W + x;
  if (out_u8) out_u8[o] = (unsigned char)val;
  if (out_f) out_f[o] = (val / 255.f - 0.5f) * 2.f;                    // load_X_one_proc scaling (utils.py:340-342)
}

// waves [N][5], nodes [N][7][8], nnode [N] (device).  out_f (or NULL): network input [N][H][W][1] in [-1,1];
// out_u8 (or NULL): the uint8 frame as it would be written to a PNG.  noise = 0: canvas only (tests).
extern "C" int spnet_fake_espi(const float* waves, const float* nodes, const int* nnode, int N, int H, int W,
                               unsigned seed, int noise, float* out_f, unsigned char* out_u8, void* stream) {
  if (N < 1 || H < 1 || W < 1 || (!out_f && !out_u8)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(fake_espi_kernel, dim3((W + 255) / 256, H, N), dim3(256), 0, (hipStream_t)stream, waves, nodes,
                     nnode, H, W, seed, noise, out_f, out_u8);
  SPNET_RETURN_LAUNCH_STATUS();
}
