// fabric_probe.hip -- what the MI355X memory side delivers for the step kernel's two traffic classes, alone and
// together, with no env arithmetic at all ("the saturated resource"):
//   stream : per env 28 B read (12 + 12 + 4, three coalesced streams) and 133 B written (116 B observation row with
//            non-temporal 16-B stores, 12 + 4 + 1 B state / reward / done) -- the step's compulsory bytes
//   gather : per env one random 128-B line of a 19 MB table (8 lanes x 16 B, like the coefficient-row gather) and one
//            128-B line of a 1 MB table (the day slice), reduced to one float so the loads cannot be dropped
//   both   : one kernel doing both per env (what k_step64 does, minus the arithmetic)
// Build + run (no torch):  hipcc --offload-arch=gfx950 -O3 tools/fabric_probe.hip -o /tmp/fabric_probe && /tmp/fabric_probe
// -DPROBE_LIB: no main(); extern "C" entry points for bench.py, which runs the probe IN ITS OWN PROCESS on the tables and
// the episode tuples of the env it is about to time (weather2alert_amd/build.py: build_probe_lib -> _lib/libw2a_probe.so),
// so that every bench line carries this box's own copy rate and probe time next to the kernel's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
#ifdef PROBE_PACKED  // what-if: the per-env state streams as 8 + 8 B in and 8 B out instead of 12 + 12 in, 12 out
struct u3 { uint32_t b, c; uint32_t a_() const { return 0; } };
#define U3_A(v) 0u
#else
struct u3 { uint32_t a, b, c; };
#define U3_A(v) (v).a
#endif

#ifdef PROBE_ZSTREAM  // what-if: no coefficient-row gather; the baseline logit's table part streams in as 4 B per env from a
                      // per-episode [day][env] array and the four run-time coefficients as 16 B per env
__device__ const float *g_Z;
__device__ const float4 *g_Wrt;
#endif
#ifdef PROBE_TABLE  // what-if: no coefficient-row gather. The table-sourced part of both logits comes as an
                    // f32 pair from a day slice L[day][feature row][draw][2] (PROBE_TABLE=1; =2: one f32 per head from
                    // two slices, the effectiveness one read by ~5 % of the envs) and the 4 run-time coefficients of both
                    // heads from a compact [column * draws][8] f32 table (2.3 MB), both by lane = env
__device__ const float *g_L;      // [153][R * 100][2]  (or two [153][R*100] planes)
__device__ const float4 *g_W8;    // [S * 100][2] float4
#define PT_DRAWS 100u
#define PT_Y 11u
#endif
#ifdef PROBE_ONEPASS  // what-if: all 64 envs of the wave gathered in ONE pass of 8 rounds (two memory hops per wave
                      // instead of three, twice the gathered bytes in flight), one 64-env observation flush
#define P_PASSES 1
#define P_ROUNDS 8
#else
#define P_PASSES 2
#define P_ROUNDS 4
#endif
#define P_PASS_ENVS (P_ROUNDS * 8)
#ifdef PROBE_LIB
__constant__ uint32_t g_rows_per_day;  // feature rows per day slice of the caller's table
#endif
template <bool STREAM, bool GATHER>
__global__ __launch_bounds__(256, 4) void k_probe(const u3 *hot, const u3 *stepc, const int32_t *act, u3 *hot_out,
                                                  float *reward, uint8_t *done, float *obs, const float4 *W,
                                                  const float4 *X, const uint32_t *wrow, const uint32_t *xrow, int64_t n) {
  const int64_t n_raw = n;
#ifdef PROBE_DAYS
#ifdef PROBE_LIB
  X += (size_t)(n >> 40) * g_rows_per_day * 8;
#else
  X += (size_t)(n >> 40) * 8206 * 8;  // day slice in the upper bits of n (keeps the signature)
#endif
  n &= (1ll << 40) - 1;
#endif
  __shared__ float tile[4][P_PASS_ENVS * 32];
#ifdef PROBE_LDS_PAD   // caps the occupancy like k_step64's register count does (-DPROBE_LDS_PAD=bytes)
  __shared__ float pad[PROBE_LDS_PAD / 4];
  if (n < 0) pad[threadIdx.x] = 0.f;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t env0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  if (env0 >= n) return;
  const uint32_t e = (uint32_t)(env0 + lane);
  float acc = 0.0f;
  u3 h = {}, c = {};
  int32_t a = 0;
  if (STREAM) { h = hot[e]; c = stepc[e]; a = act[e]; }
#ifdef PROBE_ZSTREAM
  if (STREAM) {
    const float zz = g_Z[(size_t)(n_raw >> 40) * (size_t)n + e];
    const float4 wr = g_Wrt[e];
    acc = zz + wr.x * 0.5f + wr.y * 0.25f + wr.z * 0.125f + wr.w;
  }
#endif
#ifdef PROBE_TABLE
  if (GATHER) {
    // c.b = feature row, c.c = coefficient row (column * draws + draw); without augmentation column = county of the row
    const uint32_t smp = c.c % PT_DRAWS, rows = (uint32_t)(
#ifdef PROBE_LIB
        g_rows_per_day
#else
        8206u
#endif
        );
    const size_t day = (size_t)(n_raw >> 40);
    const uint32_t li = c.b * PT_DRAWS + smp, wi = (c.b / PT_Y) * PT_DRAWS + smp;
    const bool need = ((c.c * 2654435761u) >> 24) < 13u;  // ~5 % of the envs alert today with an open gate
#if PROBE_TABLE == 3
    // ONE 16-B entry per env and head: {table part of the logit, the three run-time coefficients} from a day slice
    // [day][feature row][draw] of 16-B entries (12.7 MB per day and head at 7 920 rows x 100 draws); no other gather
    const float4 l = reinterpret_cast<const float4 *>(g_L)[day * rows * PT_DRAWS + li];
    acc += l.x + l.y * 0.5f + l.z * 0.25f + l.w * 0.125f;
    if (need) {
      const float4 e = reinterpret_cast<const float4 *>(g_L)[((size_t)153 + day) * rows * PT_DRAWS + li];
      acc += e.x + e.y * 0.5f + e.z * 0.25f + e.w * 0.125f;
    }
#elif PROBE_TABLE == 1
    const float2 l = reinterpret_cast<const float2 *>(g_L)[day * rows * PT_DRAWS + li];
    acc += l.x + (need ? l.y : 0.f);
#else
    const float lb = g_L[day * rows * PT_DRAWS + li];
    float le = 0.f;
    if (need) le = g_L[(size_t)153 * rows * PT_DRAWS + day * rows * PT_DRAWS + li];
    acc += lb + le;
#endif
#if PROBE_TABLE != 3
    const float4 wb = g_W8[wi * 2];
    float4 we = make_float4(0.f, 0.f, 0.f, 0.f);
    if (need) we = g_W8[wi * 2 + 1];
    acc += wb.x * 0.5f + wb.y * 0.25f + wb.z * 0.125f + wb.w + we.x + we.y * 0.5f + we.z * 0.25f + we.w * 0.125f;
#else
    (void)wi;
#endif
  }
#endif
  const int p = lane & 7, g = lane >> 3;
#ifdef PROBE_DEP  // like k_step64: the gather indices are part of the streamed state (lane = env) and reach the
                  // 8-lanes-per-row mapping through LDS, so the gathers wait for the state loads
  __shared__ uint2 desc[4][64];
  desc[wave][lane] = make_uint2(c.b, c.c);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
  for (int pass = 0; pass < P_PASSES; ++pass) {
    float4 x[P_ROUNDS], w[P_ROUNDS];
#pragma unroll
    for (int r = 0; r < P_ROUNDS; ++r) {
      const uint32_t j = (uint32_t)env0 + pass * P_PASS_ENVS + r * 8 + g;
      x[r] = make_float4(1.f, 2.f, 3.f, 4.f);
#ifdef PROBE_RANDOM_DATA  // stream-only mode too writes values that do not repeat
      {
        uint32_t q = (j * 8u + (uint32_t)p) * 2654435761u + (uint32_t)(n_raw >> 40) * 40503u;
        q ^= q >> 15; q *= 2246822519u; q ^= q >> 13;
        x[r] = make_float4(__uint_as_float(0x3F800000u | (q & 0x7FFFFFu)), __uint_as_float(0x3F800000u | ((q >> 3) & 0x7FFFFFu)),
                           __uint_as_float(0x3F800000u | ((q >> 6) & 0x7FFFFFu)), __uint_as_float(0x3F800000u | ((q >> 9) & 0x7FFFFFu)));
      }
#endif
      w[r] = x[r];
      if (GATHER) {
#ifdef PROBE_DEP
        const uint2 dd = desc[wave][pass * P_PASS_ENVS + r * 8 + g];
        x[r] = X[dd.x * 8 + p];
#if defined(PROBE_ZSTREAM) || defined(PROBE_TABLE)
        w[r] = x[r];
#elif defined(PROBE_WDENSE)  // what-if: baseline coefficient rows on consecutive 128-B lines ([head][row][32] instead of
                             // [row][head][32], where they sit on every other line: half the L2 channels / sets)
        w[r] = W[dd.y * 8 + p];
#else
        w[r] = W[dd.y * 16 + p];
#endif
#else
        x[r] = X[xrow[j] * 8 + p];  // xrow / wrow are read as 8-lane broadcasts (the real kernel takes them from LDS)
        w[r] = W[wrow[j] * 16 + p];
#endif
      }
    }
#pragma unroll
    for (int r = 0; r < P_ROUNDS; ++r) {
#ifdef PROBE_F64  // the step kernel's arithmetic: two 4-term fp64 chains and an 8-lane fp64 all-reduce per env
      {
        double zb = (double)x[r].x * (double)w[r].x, ze = (double)x[r].x * (double)w[r].y;
        zb = fma((double)x[r].y, (double)w[r].y, zb); ze = fma((double)x[r].y, (double)w[r].z, ze);
        zb = fma((double)x[r].z, (double)w[r].z, zb); ze = fma((double)x[r].z, (double)w[r].w, ze);
        zb = fma((double)x[r].w, (double)w[r].w, zb); ze = fma((double)x[r].w, (double)w[r].x, ze);
        for (int m = 1; m < 8; m <<= 1) {
          zb += __shfl_xor(zb, m);
          ze += __shfl_xor(ze, m);
        }
        acc += __builtin_amdgcn_rcpf(1.0f + __expf(-(float)zb)) * __builtin_amdgcn_rcpf(1.0f + __expf(-(float)ze));
      }
#else
      acc += x[r].x * w[r].x + x[r].y * w[r].y + x[r].z * w[r].z + x[r].w * w[r].w;
#endif
      if (STREAM) {
        float *t = tile[wave] + (r * 8 + g) * 29 + p * 4;
        if (p < 7) { t[0] = x[r].x; t[1] = x[r].y; t[2] = x[r].z; t[3] = x[r].w; } else t[0] = x[r].x;
      }
    }
    if (STREAM) {
      __builtin_amdgcn_wave_barrier();
      float *dst = obs + (env0 + pass * P_PASS_ENVS) * 29;
      for (int c0 = 0; c0 < P_PASS_ENVS * 8; c0 += 64) {
        const int ch = c0 + lane;
        if (ch < P_PASS_ENVS * 29 / 4) __builtin_nontemporal_store(reinterpret_cast<const v4f *>(tile[wave])[ch], reinterpret_cast<v4f *>(dst) + ch);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // lane = env outputs
  acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
  if (STREAM) {
#ifdef PROBE_PACKED
    h.b += (uint32_t)a; h.c = __float_as_uint(acc + (float)c.b);
#else
    h.a += (uint32_t)a; h.c = __float_as_uint(acc + __uint_as_float(c.a) + (float)c.b);
#endif
    hot_out[e] = h;
    reward[e] = acc;
    done[e] = (uint8_t)(h.b & 1u);
  } else if (acc == 123456.789f) {
    reward[e] = acc;  // keeps the gathers alive without writing
  }
}

#ifdef PROBE_LIB
// ------------------------------------------------------------------------------------------------------------------
// In-process entry points (C ABI, plain pointers). Nothing here is part of the env: measurement only.
// float4 copy, ONE 16-B element per thread and as many workgroups as that takes: of the forms tried on MI355X (grid-stride
// loops of 1 024 .. 16 384 workgroups with 4 or 8 loads in flight, per-workgroup chunks, non-temporal stores,
// hipMemcpyAsync: 4.4 .. 5.9 TB/s, profiles/r04/copy_explore.log) the only one that reaches the 6.2-6.3 TB/s
// MI355X_MICROARCH.md quotes for "float4 copy"
__global__ __launch_bounds__(256) void k_copy16(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
// every device / host allocation of an entry point is registered here and released on EVERY way out (a failing second
// hipMalloc used to leak the first GiB into the process that goes on to run the bench's extras)
struct Scratch {
  void *dev[16]; int nd = 0; void *host[4]; int nh = 0; hipEvent_t ev[2]; int ne = 0;
  ~Scratch() {
    for (int i = 0; i < ne; ++i) (void)hipEventDestroy(ev[i]);
    for (int i = 0; i < nd; ++i) (void)hipFree(dev[i]);
    for (int i = 0; i < nh; ++i) free(host[i]);
  }
  template <class T> hipError_t alloc(T **p, size_t bytes) { hipError_t e = hipMalloc((void **)p, bytes); if (e == hipSuccess) dev[nd++] = *p; return e; }
  template <class T> T *halloc(size_t bytes) { T *p = (T *)malloc(bytes); if (p) host[nh++] = p; return p; }
  hipError_t event(hipEvent_t *e) { hipError_t r = hipEventCreate(e); if (r == hipSuccess) ev[ne++] = *e; return r; }
};
#define TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(g_perr, sizeof g_perr, "%s: %s", #x, hipGetErrorString(e_)); return -3; } } while (0)
static char g_perr[256] = "";
extern "C" const char *w2a_probe_last_error(void) { return g_perr; }

// copy of `bytes` (read + write = 2 x bytes through the fabric), best of `reps` per variant: GB/s of this box right
// now. gbs_out[0] = the best of gbs_out[1..2] = {float4 kernel, hipMemcpyAsync device-to-device}
extern "C" int w2a_probe_copy(size_t bytes, int reps, float *gbs_out, void *stream) {
  if (!gbs_out || bytes < (1u << 20) || reps <= 0) { snprintf(g_perr, sizeof g_perr, "w2a_probe_copy: bad argument"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  v4f *a = nullptr, *b = nullptr;
  TRY(sc.alloc(&a, bytes)); TRY(sc.alloc(&b, bytes));
  TRY(hipMemsetAsync(a, 0x3c, bytes, s));
  hipEvent_t e0, e1; TRY(sc.event(&e0)); TRY(sc.event(&e1));
  gbs_out[0] = 0.0f;
  for (int variant = 0; variant < 2; ++variant) {
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
      TRY(hipEventRecord(e0, s));
      if (variant == 0) hipLaunchKernelGGL(k_copy16, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, s, a, b, bytes / 16);
      else TRY(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, s));
      TRY(hipEventRecord(e1, s)); TRY(hipEventSynchronize(e1));
      float ms; TRY(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) best = ms;
    }
    gbs_out[1 + variant] = (float)(2.0 * (double)bytes / (best * 1e-3) / 1e9);
    if (gbs_out[1 + variant] > gbs_out[0]) gbs_out[0] = gbs_out[1 + variant];
  }
  TRY(hipStreamSynchronize(s));
  return 0;
}

// The step's traffic without any env logic on the CALLER's tables and index pattern: X [T][rows_per_day][8] float4,
// W [w_rows][16] float4, xrow / wrow [n] = feature row and coefficient row of every env (n a multiple of 256).
// us_out[0..2] = microseconds per launch (back to back, best of `reps` runs of `iters` launches): streams only, gathers
// only, both. Scratch (state words, outputs) is allocated and freed here.
extern "C" int w2a_probe_step_pattern(const void *X, uint32_t rows_per_day, int32_t T, const void *W, const uint32_t *xrow,
                                      const uint32_t *wrow, int64_t n, int reps, int iters, float *us_out, void *stream) {
  if (!X || !W || !xrow || !wrow || !us_out || n <= 0 || (n & 255) || T <= 0 || reps <= 0 || iters <= 0) {
    snprintf(g_perr, sizeof g_perr, "w2a_probe_step_pattern: bad argument (n must be a multiple of 256)");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  u3 *hot = nullptr, *stepc = nullptr; int32_t *act = nullptr; float *reward = nullptr, *obs = nullptr; uint8_t *done = nullptr;
  TRY(sc.alloc(&hot, n * sizeof(u3))); TRY(sc.alloc(&stepc, n * sizeof(u3))); TRY(sc.alloc(&act, n * 4));
  TRY(sc.alloc(&reward, n * 4)); TRY(sc.alloc(&done, n)); TRY(sc.alloc(&obs, n * 29 * 4));
  TRY(hipMemsetAsync(hot, 0, n * sizeof(u3), s)); TRY(hipMemsetAsync(act, 0, n * 4, s));
  TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rows_per_day), &rows_per_day, sizeof(rows_per_day), 0, hipMemcpyHostToDevice, s));
  {  // stepc.b / .c carry the gather indices (PROBE_DEP: they reach the gathers through the streamed state, like the env's)
    u3 *hs = sc.halloc<u3>(n * sizeof(u3));
    uint32_t *hx = sc.halloc<uint32_t>(n * 4), *hw = sc.halloc<uint32_t>(n * 4);
    if (!hs || !hx || !hw) { snprintf(g_perr, sizeof g_perr, "w2a_probe_step_pattern: out of host memory"); return -2; }
    TRY(hipMemcpyAsync(hx, xrow, n * 4, hipMemcpyDeviceToHost, s)); TRY(hipMemcpyAsync(hw, wrow, n * 4, hipMemcpyDeviceToHost, s));
    TRY(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i) { hs[i] = u3{}; hs[i].b = hx[i]; hs[i].c = hw[i]; }
    TRY(hipMemcpyAsync(stepc, hs, n * sizeof(u3), hipMemcpyHostToDevice, s)); TRY(hipStreamSynchronize(s));
  }
  hipEvent_t e0, e1; TRY(sc.event(&e0)); TRY(sc.event(&e1));
  const dim3 grid((unsigned)(n / 256)), block(256);
  const float4 *Xp = (const float4 *)X, *Wp = (const float4 *)W;
  for (int k = 0; k < 3; ++k) {
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
      TRY(hipEventRecord(e0, s));
      for (int it = 0; it < iters; ++it) {
        const int64_t nn = ((int64_t)(it % T) << 40) | n;  // PROBE_DAYS: another day slice every launch
        if (k == 0) hipLaunchKernelGGL((k_probe<true, false>), grid, block, 0, s, hot, stepc, act, hot, reward, done, obs, Wp, Xp, wrow, xrow, nn);
        if (k == 1) hipLaunchKernelGGL((k_probe<false, true>), grid, block, 0, s, hot, stepc, act, hot, reward, done, obs, Wp, Xp, wrow, xrow, nn);
        if (k == 2) hipLaunchKernelGGL((k_probe<true, true>), grid, block, 0, s, hot, stepc, act, hot, reward, done, obs, Wp, Xp, wrow, xrow, nn);
      }
      TRY(hipEventRecord(e1, s)); TRY(hipEventSynchronize(e1));
      float ms; TRY(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    us_out[k] = best * 1e3f / (float)iters;
  }
  TRY(hipGetLastError());
  TRY(hipStreamSynchronize(s));
  return 0;
}
#else
int main() {
  const int64_t n = 1 << 20;
  const int S = 74600, R = 8206;  // coefficient rows (x 256 B) and day-slice rows (x 128 B)
  u3 *hot, *stepc, *hot_out; int32_t *act; float *reward, *obs; uint8_t *done; float4 *W, *X; uint32_t *wrow, *xrow;
  CHECK(hipMalloc(&hot, n * sizeof(u3))); CHECK(hipMalloc(&stepc, n * sizeof(u3))); CHECK(hipMalloc(&hot_out, n * sizeof(u3)));
  CHECK(hipMalloc(&act, n * 4)); CHECK(hipMalloc(&reward, n * 4)); CHECK(hipMalloc(&done, n));
#ifdef PROBE_UC_OBS  // what-if: the observation buffer in memory the L2 does not allocate for (1 = fine-grained, 3 = uncached): the
                     // 121 MB of rows per launch would stop evicting the table lines the gathers want to find in L2
  CHECK(hipExtMallocWithFlags((void **)&obs, n * 29 * 4, PROBE_UC_OBS));
#else
  CHECK(hipMalloc(&obs, n * 29 * 4));
#endif
  CHECK(hipMalloc(&W, (size_t)S * 256)); CHECK(hipMalloc(&X, (size_t)R * 128 * 153));
  CHECK(hipMalloc(&wrow, n * 4)); CHECK(hipMalloc(&xrow, n * 4));
  CHECK(hipMemset(hot, 0, n * sizeof(u3))); CHECK(hipMemset(stepc, 0, n * sizeof(u3))); CHECK(hipMemset(act, 0, n * 4));
  CHECK(hipMemset(W, 0, (size_t)S * 256)); CHECK(hipMemset(X, 0, (size_t)R * 128 * 153));
#ifdef PROBE_RANDOM_DATA  // random table values instead of zeros (arithmetic on zeros draws less power: clocks differ)
  {
    const size_t nx = (size_t)R * 32 * 153, nw = (size_t)S * 64;
    float *hb = (float *)malloc((nx > nw ? nx : nw) * 4);
    uint64_t q = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < nx; ++i) { q ^= q << 13; q ^= q >> 7; q ^= q << 17; hb[i] = (float)((q >> 40) * (1.0 / 16777216.0)); }
    CHECK(hipMemcpy(X, hb, nx * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nw; ++i) { q ^= q << 13; q ^= q >> 7; q ^= q << 17; hb[i] = (float)((q >> 40) * (1.0 / 16777216.0)) - 0.5f; }
    CHECK(hipMemcpy(W, hb, nw * 4, hipMemcpyHostToDevice));
    free(hb);
  }
#endif
#ifdef PROBE_ZSTREAM
  {
    float *Z; float4 *Wrt;
    CHECK(hipMalloc(&Z, (size_t)153 * n * 4)); CHECK(hipMalloc(&Wrt, n * 16));
    CHECK(hipMemset(Z, 0x3c, (size_t)153 * n * 4)); CHECK(hipMemset(Wrt, 0x3c, n * 16));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_Z), &Z, sizeof(Z))); CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_Wrt), &Wrt, sizeof(Wrt)));
  }
#endif
#ifdef PROBE_TABLE
  {
    float *L; float4 *W8;
    const size_t lbytes = (size_t)153 * R * 100 * (PROBE_TABLE == 3 ? 32 : 8);  // =3: 16 B per head
    CHECK(hipMalloc(&L, lbytes)); CHECK(hipMalloc(&W8, (size_t)S * 32));
    CHECK(hipMemset(L, 0x3c, lbytes)); CHECK(hipMemset(W8, 0x3c, (size_t)S * 32));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_L), &L, sizeof(L))); CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_W8), &W8, sizeof(W8)));
  }
#endif
  uint32_t *hw = (uint32_t *)malloc(n * 4), *hx = (uint32_t *)malloc(n * 4);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17; hw[i] = (uint32_t)(s % S);
    s ^= s << 13; s ^= s >> 7; s ^= s << 17; hx[i] = (uint32_t)(s % R);
  }
  for (int mode = 0; mode < 2; ++mode) {  // 0: uniform rows (configs[3]-like), 1: rows of the first 25 000 only (configs[2]-like)
    if (mode == 1) for (int64_t i = 0; i < n; ++i) hw[i] %= 25000;
    CHECK(hipMemcpy(wrow, hw, n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(xrow, hx, n * 4, hipMemcpyHostToDevice));
    {
      u3 *hs = (u3 *)malloc(n * sizeof(u3));
      for (int64_t i = 0; i < n; ++i) { hs[i] = u3{}; hs[i].b = hx[i]; hs[i].c = hw[i]; }
      CHECK(hipMemcpy(stepc, hs, n * sizeof(u3), hipMemcpyHostToDevice));
      free(hs);
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid((unsigned)(n / 256)), block(256);
    const char *names[3] = {sizeof(u3) == 8 ? "stream only (149 B/env: 8 + 8 + 4 in, 8 + 121 out)" : "stream only (161 B/env compulsory)", "gather only (2 x 128-B lines/env)", "stream + gather"};
    for (int k = 0; k < 3; ++k) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 100; ++it) {
#ifdef PROBE_DAYS
          const int64_t n = ((int64_t)(it % 153) << 40) | (1 << 20);
#endif
#ifdef PROBE_INPLACE  // like the env: the state words are updated in place
          u3 *hin = hot, *hout = hot;
#else
          u3 *hin = (it & 1) ? hot_out : hot, *hout = (it & 1) ? hot : hot_out;
#endif
          if (k == 0) hipLaunchKernelGGL((k_probe<true, false>), grid, block, 0, 0, hin, stepc, act, hout, reward, done, obs, W, X, wrow, xrow, n);
          if (k == 1) hipLaunchKernelGGL((k_probe<false, true>), grid, block, 0, 0, hin, stepc, act, hout, reward, done, obs, W, X, wrow, xrow, n);
          if (k == 2) hipLaunchKernelGGL((k_probe<true, true>), grid, block, 0, 0, hin, stepc, act, hout, reward, done, obs, W, X, wrow, xrow, n);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%s %-36s: %6.2f us per launch (back to back, incl. the launch boundary)\n",
             mode ? "[rows < 25 000]" : "[uniform rows ]", names[k], best * 10.0f);
    }
  }
  return 0;
}
#endif  // PROBE_LIB
