// Standalone A/B harness for the two lattice layer-1 GEMMs (no Python, no torch: a GPU visit costs seconds, not the minutes
// of an interpreter start).  Loads the product library, builds the packed operands with the library's own producers, then runs
// the forward / backward entry point under a list of ENVIRONMENT configurations (the launchers read their knobs at every call) and
// reports the time per launch and whether the outputs are BIT-IDENTICAL to those of the first configuration.
//
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lat_bench.hip -o tools/micro/lat_bench -ldl
//   tools/micro/lat_bench LIB.so fwd|bwd|both  "RCMARL_LAT_PERSIST=0" "RCMARL_LAT_PERSIST=1,RCMARL_LAT_STAGGER=4" ...
// env: LB_S (16) LB_N (256) LB_B (3000) LB_ITERS (20) LB_WIDTHS ("2,3")
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef int (*encode_t)(const float*, long, const float*, int, int, int, void*, int, int, void*, int, int, int*, void*);
typedef int (*split_t)(const float*, const float*, void*, int, int, int, int, int, int, int, void*);
typedef int (*packdz_t)(const float*, void*, int, int, int, int, int, int, int, void*);
typedef int (*fwd_t)(const void*, int, int, const void*, int, int, const float*, float*, int, int, int, int, int, int, int, void*);
typedef int (*bwd_t)(const void*, int, int, const void*, int, int, const float*, float*, const int*, int, int, int, int, int, int, float,
                     void*, int, int, void*);
typedef int (*bwdws_t)(const void*, int, int, const void*, int, int, const float*, float*, const int*, int, int, int, int, int, int, float,
                       void*, int, int, void*, long, void*);
typedef long (*wsbytes_t)(int, int, int, int, int);

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void k_init_x(float* x, long n, int in_dim, float alpha, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int pos = (int)(hash32((unsigned)i * 2654435761u + seed) & 31);
    x[i] = alpha * (float)(2 * pos - 31);
  }
}
__global__ void k_init_uniform(float* x, long n, float scale, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] = scale * ((float)(hash32((unsigned)i * 2246822519u + seed) >> 8) * (1.f / 8388608.f) - 1.f);
}
__global__ void k_fill(float* x, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void k_diff(const unsigned* a, const unsigned* b, long n, unsigned long long* count) {
  unsigned long long c = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(count, c);
}

static long ndiff(const void* a, const void* b, long bytes, unsigned long long* d_count) {
  CK(hipMemset(d_count, 0, 8));
  hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, (const unsigned*)a, (const unsigned*)b, bytes / 4, d_count);
  unsigned long long h = 0;
  CK(hipMemcpy(&h, d_count, 8, hipMemcpyDeviceToHost));
  return (long)h;
}

static int envi(const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; }
static std::vector<std::pair<std::string, std::string>> parse_cfg(const std::string& s) {
  std::vector<std::pair<std::string, std::string>> kv;
  size_t p = 0;
  while (p < s.size()) {
    size_t c = s.find(',', p);
    if (c == std::string::npos) c = s.size();
    const std::string item = s.substr(p, c - p);
    const size_t eq = item.find('=');
    if (eq != std::string::npos) kv.push_back({item.substr(0, eq), item.substr(eq + 1)});
    p = c + 1;
  }
  return kv;
}
static int cdiv(int a, int b) { return (a + b - 1) / b; }
static int pad64(int a) { return (a + 63) / 64 * 64; }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s LIB.so fwd|bwd|both CONFIG...\n", argv[0]); return 1; }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  const std::string what = argv[2];
  encode_t enc = (encode_t)dlsym(h, "rcmarl_lattice_encode");
  split_t split = (split_t)dlsym(h, "rcmarl_w1_split");
  packdz_t packdz = (packdz_t)dlsym(h, "rcmarl_lattice_pack_dz");
  fwd_t fwd0 = (fwd_t)dlsym(h, "rcmarl_layer1_forward_lattice");
  bwd_t bwd0 = (bwd_t)dlsym(h, "rcmarl_layer1_backward_sgd_lattice");
  bwdws_t bwdws0 = (bwdws_t)dlsym(h, "rcmarl_layer1_backward_sgd_lattice_ws");
  wsbytes_t wsbytes = (wsbytes_t)dlsym(h, "rcmarl_lattice_backward_workspace_bytes");
  if (!enc || !split || !packdz || !fwd0 || !bwd0) { fprintf(stderr, "missing symbols\n"); return 1; }
  const int S = envi("LB_S", 16), N = envi("LB_N", 256), B = envi("LB_B", 3000), HID = 20, iters = envi("LB_ITERS", 20);
  const char* widths = getenv("LB_WIDTHS") ? getenv("LB_WIDTHS") : "2,3";
  unsigned long long* d_count;
  CK(hipMalloc(&d_count, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d MHz; S=%d N=%d B=%d iters=%d\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, S, N, B, iters);

  for (const char* wp_ = widths; *wp_;) {
    const int width = atoi(wp_);
    while (*wp_ && *wp_ != ',') ++wp_;
    if (*wp_ == ',') ++wp_;
    const int in_dim = width * N;
    const int P = in_dim * HID + HID + HID * HID + HID + HID + 1, ldp = pad64(P), ldb = pad64(B);
    const int b_pad = cdiv(B, 256) * 256;
    const int kp_rt = b_pad / 128, kp_kt = cdiv(in_dim, 32), ktp_rt = 2 * cdiv(in_dim, 256), ktp_kt = b_pad / 32;
    const int wp_rt = cdiv(N * HID, 128), wp_kt = cdiv(in_dim, 32), dzp_rt = cdiv(N * HID, 128), dzp_kt = b_pad / 32;
    const long PK = 8192;
    const long n_x = (long)S * B * in_dim, n_th = (long)S * N * ldp, n_a1 = (long)S * N * HID * ldb;
    const long kp_b = (long)S * kp_rt * kp_kt * PK, ktp_b = (long)S * ktp_rt * ktp_kt * PK, wp_b = (long)S * wp_rt * wp_kt * 3 * PK,
               dzp_b = (long)S * dzp_rt * dzp_kt * 3 * PK;
    float *x, *alpha, *theta0, *theta, *theta_ref, *a1t, *a1t_ref, *dz;
    unsigned char *kp, *ktp, *wp, *dzp, *wp_out, *wp_ref;
    int *flag, *mask;
    CK(hipMalloc(&x, n_x * 4)); CK(hipMalloc(&alpha, in_dim * 4)); CK(hipMalloc(&theta0, n_th * 4)); CK(hipMalloc(&theta, n_th * 4));
    CK(hipMalloc(&theta_ref, n_th * 4)); CK(hipMalloc(&a1t, n_a1 * 4)); CK(hipMalloc(&a1t_ref, n_a1 * 4)); CK(hipMalloc(&dz, n_a1 * 4));
    CK(hipMalloc(&kp, kp_b)); CK(hipMalloc(&ktp, ktp_b)); CK(hipMalloc(&wp, wp_b)); CK(hipMalloc(&dzp, dzp_b));
    CK(hipMalloc(&wp_out, wp_b)); CK(hipMalloc(&wp_ref, wp_b)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&mask, N * 4));
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(kp, 0, kp_b)); CK(hipMemset(ktp, 0, ktp_b)); CK(hipMemset(wp, 0, wp_b));
    CK(hipMemset(dzp, 0, dzp_b)); CK(hipMemset(wp_out, 0, wp_b)); CK(hipMemset(wp_ref, 0, wp_b));
    CK(hipMemset(a1t, 0, n_a1 * 4)); CK(hipMemset(a1t_ref, 0, n_a1 * 4));
    {
      std::vector<int> m(N, 1);
      m[N / 3] = 0;                                          // one agent masked out of the update (adversary rows)
      CK(hipMemcpy(mask, m.data(), N * 4, hipMemcpyHostToDevice));
    }
    const float al = 0.5f / 9.2330384f;                       // 0.5 / std(arange(32))
    hipLaunchKernelGGL(k_init_x, dim3(4096), dim3(256), 0, 0, x, n_x, in_dim, al, 12345u + width);
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, alpha, (long)in_dim, al);
    hipLaunchKernelGGL(k_init_uniform, dim3(4096), dim3(256), 0, 0, theta0, n_th, 0.08f, 777u + width);
    hipLaunchKernelGGL(k_init_uniform, dim3(4096), dim3(256), 0, 0, dz, n_a1, 2e-3f, 999u + width);
    int rc = enc(x, (long)B * in_dim, alpha, S, B, in_dim, kp, kp_rt, kp_kt, ktp, ktp_rt, ktp_kt, flag, nullptr);
    rc |= split(theta0, alpha, wp, S, N, in_dim, HID, ldp, wp_rt, wp_kt, nullptr);
    rc |= packdz(dz, dzp, S, N, B, HID, ldb, dzp_rt, dzp_kt, nullptr);
    CK(hipDeviceSynchronize());
    int hflag = 0;
    CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
    printf("== in_dim %d: producers rc=%d lattice-flag=%d\n", in_dim, rc, hflag);
    long ws_bytes = wsbytes ? wsbytes(S, N, B, in_dim, HID) : 0;
    void* ws = nullptr;
    if (ws_bytes > 0) { CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes)); }
    const double flops = 2.0 * S * N * HID * (double)B * in_dim;

    for (int ci = 3; ci < argc; ++ci) {
      const auto kv = parse_cfg(argv[ci]);
      for (auto& p : kv) setenv(p.first.c_str(), p.second.c_str(), 1);
      fwd_t fwd = fwd0; bwd_t bwd = bwd0; bwdws_t bwdws = bwdws0;
      void* hv = nullptr;
      if (getenv("LIB")) {                                    // a variant build of the library for this configuration's GEMM calls
        hv = dlopen(getenv("LIB"), RTLD_NOW | RTLD_LOCAL);
        if (!hv) { fprintf(stderr, "dlopen %s: %s\n", getenv("LIB"), dlerror()); return 1; }
        fwd = (fwd_t)dlsym(hv, "rcmarl_layer1_forward_lattice");
        bwd = (bwd_t)dlsym(hv, "rcmarl_layer1_backward_sgd_lattice");
        bwdws = (bwdws_t)dlsym(hv, "rcmarl_layer1_backward_sgd_lattice_ws");
      }
      typedef void (*tsdump_t)(const char*);
      tsdump_t tsdump = (tsdump_t)dlsym(hv ? hv : h, "rcmarl_lat_ts_dump");
      const bool use_ws = getenv("LB_WS") && atoi(getenv("LB_WS")) && bwdws && ws;
      if (what == "fwd" || what == "both") {
        CK(hipMemset(a1t, 0xff, n_a1 * 4));
        rc = fwd(kp, kp_rt, kp_kt, wp, wp_rt, wp_kt, theta0, a1t, S, N, B, in_dim, HID, ldp, ldb, nullptr);
        CK(hipDeviceSynchronize());
        long nd = -1;
        if (ci == 3) CK(hipMemcpy(a1t_ref, a1t, n_a1 * 4, hipMemcpyDeviceToDevice)); else nd = ndiff(a1t, a1t_ref, n_a1 * 4, d_count);
        for (int i = 0; i < 3; ++i) fwd(kp, kp_rt, kp_kt, wp, wp_rt, wp_kt, theta0, a1t, S, N, B, in_dim, HID, ldp, ldb, nullptr);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) fwd(kp, kp_rt, kp_kt, wp, wp_rt, wp_kt, theta0, a1t, S, N, B, in_dim, HID, ldp, ldb, nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("fwd in=%4d %-70s rc=%d %8.1f us %6.1f TF/s fp32-eq  diff-vs-first=%ld\n", in_dim, argv[ci], rc, us, flops / us / 1e6, nd);
        if (tsdump) { char tag[64]; snprintf(tag, sizeof tag, "fwd in=%d cfg %d", in_dim, ci - 3); tsdump(tag); }
      }
      if (what == "bwd" || what == "both") {
        void* wp_arg = (getenv("LB_NOWP") && atoi(getenv("LB_NOWP"))) ? nullptr : (void*)wp_out;      // knock-out: no next-forward operand
        auto call = [&](float lr) {
          return use_ws ? bwdws(ktp, ktp_rt, ktp_kt, dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, HID, ldp, lr, wp_arg, wp_rt,
                                wp_kt, ws, ws_bytes, nullptr)
                        : bwd(ktp, ktp_rt, ktp_kt, dzp, dzp_rt, dzp_kt, alpha, theta, mask, S, N, B, in_dim, HID, ldp, lr, wp_arg, wp_rt,
                              wp_kt, nullptr);
        };
        CK(hipMemcpy(theta, theta0, n_th * 4, hipMemcpyDeviceToDevice));
        CK(hipMemset(wp_out, 0, wp_b));
        rc = call(0.01f);
        CK(hipDeviceSynchronize());
        long nd1 = -1, nd2 = -1;
        if (ci == 3) { CK(hipMemcpy(theta_ref, theta, n_th * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(wp_ref, wp_out, wp_b, hipMemcpyDeviceToDevice)); }
        else { nd1 = ndiff(theta, theta_ref, n_th * 4, d_count); nd2 = ndiff(wp_out, wp_ref, wp_b, d_count); }
        for (int i = 0; i < 3; ++i) call(1e-9f);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) call(1e-9f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("bwd in=%4d %-70s rc=%d %8.1f us %6.1f TF/s fp32-eq  diff theta=%ld wp=%ld%s\n", in_dim, argv[ci], rc, us, flops / us / 1e6, nd1,
               nd2, use_ws ? " (ws)" : "");
        if (tsdump) { char tag[64]; snprintf(tag, sizeof tag, "bwd in=%d cfg %d", in_dim, ci - 3); tsdump(tag); }
      }
      for (auto& p : kv) unsetenv(p.first.c_str());
      if (hv) dlclose(hv);
      fflush(stdout);
    }
    hipFree(x); hipFree(alpha); hipFree(theta0); hipFree(theta); hipFree(theta_ref); hipFree(a1t); hipFree(a1t_ref); hipFree(dz);
    hipFree(kp); hipFree(ktp); hipFree(wp); hipFree(dzp); hipFree(wp_out); hipFree(wp_ref); hipFree(flag); hipFree(mask);
    if (ws) hipFree(ws);
  }
  return 0;
}
