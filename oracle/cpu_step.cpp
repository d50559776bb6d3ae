// CPU ORACLE #2 -- TEST INFRASTRUCTURE AND REPORTED CPU BASELINE ONLY.  Never linked into the product
// (visdial_amd/); only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg load it.
//
// PARITY UNPINNED (as oracle/visdial_oracle.py): the reference ships no golden vectors and cannot run here
// (Lua/Torch7 absent).  This file is a THIRD, structurally different restatement of the same training step,
// written from the reference's Lua files, so that a shared misreading in the numpy oracle has an independent
// check, and so that the CPU baseline is timed on the identical B = 20 batch:
// C++17 + OpenMP, fp32 throughout (the reference's arithmetic), organised the way Torch7 runs it on the CPU
// (`th train.lua -gpuid -1`, train.lua:22-24):
//   * one pair of GEMMs per LSTM timestep (x_t*Wx, h_{t-1}*Wh) followed by pointwise gate math -- nothing is
//     hoisted out of the recurrence, the option LSTM's input projection is NOT replaced by a table gather;
//   * the image features are replicated 10x per dialog (model.lua:262-265) and pushed through the two image
//     Linears at [N*196 x 512];
//   * the 100 option clones of decoders/disc.lua:8-15 are one batch of N*100 sequences (same math).
// Scope: encoder mn-att-ques-im-hist (encoders/mn-att-ques-im-hist.lua:5-115) + decoder disc
// (decoders/disc.lua:3-32) + CrossEntropyCriterion (model.lua:37-38, 326-338) + clamp / adam
// (model.lua:96-99, model_utils/optim_updates.lua:62-91).  Module semantics (nn.SeqLSTM incl. maskZero,
// LookupTableMaskZero, Linear, Dropout, MaskSoftMax): SURVEY.md Appendix A.
//
// Build: g++ -O3 -std=c++17 -fopenmp -shared -fPIC cpu_step.cpp -o _build/libvd_cpu_step.so  (oracle/Makefile)
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------------------
// fp32 GEMM: C[M x N] (+)= op(A)[M x K] * op(B)[K x N], operands addressed through (row, col) strides so one
// routine serves NN / NT / TN.  Blocked + packed, 6 x 16 (AVX2/FMA) or 12 x 32 (AVX-512) register tiles,
// OpenMP over the C tiles.  Plain summation order per C element: k ascending in KC blocks.
// ------------------------------------------------------------------------------------------------------
constexpr int KC = 256;

typedef void (*ukr_t)(int kc, const float* Ap, const float* Bp, float* C, long ldc, int mr, int nr, bool acc);

template <int MR, int NR>
void ukr_scalar(int kc, const float* Ap, const float* Bp, float* C, long ldc, int mr, int nr, bool acc) {
  float c[MR][NR];
  for (int i = 0; i < MR; ++i)
    for (int j = 0; j < NR; ++j) c[i][j] = 0.f;
  for (int k = 0; k < kc; ++k)
    for (int i = 0; i < MR; ++i) {
      const float a = Ap[k * MR + i];
      for (int j = 0; j < NR; ++j) c[i][j] += a * Bp[k * NR + j];
    }
  for (int i = 0; i < mr; ++i)
    for (int j = 0; j < nr; ++j) C[i * ldc + j] = (acc ? C[i * ldc + j] : 0.f) + c[i][j];
}

#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) void ukr_avx2_6x16(int kc, const float* Ap, const float* Bp, float* C, long ldc,
                                                         int mr, int nr, bool acc) {
  __m256 c[6][2];
  for (int i = 0; i < 6; ++i) c[i][0] = c[i][1] = _mm256_setzero_ps();
  for (int k = 0; k < kc; ++k) {
    const __m256 b0 = _mm256_loadu_ps(Bp + k * 16), b1 = _mm256_loadu_ps(Bp + k * 16 + 8);
    for (int i = 0; i < 6; ++i) {
      const __m256 a = _mm256_broadcast_ss(Ap + k * 6 + i);
      c[i][0] = _mm256_fmadd_ps(a, b0, c[i][0]);
      c[i][1] = _mm256_fmadd_ps(a, b1, c[i][1]);
    }
  }
  float t[6][16];
  for (int i = 0; i < 6; ++i) {
    _mm256_storeu_ps(t[i], c[i][0]);
    _mm256_storeu_ps(t[i] + 8, c[i][1]);
  }
  for (int i = 0; i < mr; ++i)
    for (int j = 0; j < nr; ++j) C[i * ldc + j] = (acc ? C[i * ldc + j] : 0.f) + t[i][j];
}

__attribute__((target("avx512f"))) void ukr_avx512_12x32(int kc, const float* Ap, const float* Bp, float* C, long ldc,
                                                          int mr, int nr, bool acc) {
  __m512 c[12][2];
  for (int i = 0; i < 12; ++i) c[i][0] = c[i][1] = _mm512_setzero_ps();
  for (int k = 0; k < kc; ++k) {
    const __m512 b0 = _mm512_loadu_ps(Bp + k * 32), b1 = _mm512_loadu_ps(Bp + k * 32 + 16);
    for (int i = 0; i < 12; ++i) {
      const __m512 a = _mm512_set1_ps(Ap[k * 12 + i]);
      c[i][0] = _mm512_fmadd_ps(a, b0, c[i][0]);
      c[i][1] = _mm512_fmadd_ps(a, b1, c[i][1]);
    }
  }
  float t[12][32];
  for (int i = 0; i < 12; ++i) {
    _mm512_storeu_ps(t[i], c[i][0]);
    _mm512_storeu_ps(t[i] + 16, c[i][1]);
  }
  for (int i = 0; i < mr; ++i)
    for (int j = 0; j < nr; ++j) C[i * ldc + j] = (acc ? C[i * ldc + j] : 0.f) + t[i][j];
}
#endif

struct Kernel {
  int MR, NR;
  ukr_t fn;
  const char* name;
};

Kernel pick_kernel() {
#if defined(__x86_64__)
  const char* force = getenv("VDCPU_ISA");
  const bool want512 = !force || !strcmp(force, "avx512");
  const bool want2 = !force || !strcmp(force, "avx2") || want512;
  if (want512 && __builtin_cpu_supports("avx512f")) return Kernel{12, 32, ukr_avx512_12x32, "avx512f 12x32"};
  if (want2 && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"))
    return Kernel{6, 16, ukr_avx2_6x16, "avx2+fma 6x16"};
#endif
  return Kernel{4, 8, ukr_scalar<4, 8>, "scalar 4x8"};
}
const Kernel& kernel() {
  static const Kernel k = pick_kernel();
  return k;
}

void gemm(int M, int N, int K, const float* A, long rsa, long csa, const float* B, long rsb, long csb, float* C,
          long ldc, bool acc) {
  if (M <= 0 || N <= 0) return;
  if (K <= 0) {
    if (!acc)
      for (int i = 0; i < M; ++i) memset(C + (long)i * ldc, 0, sizeof(float) * N);
    return;
  }
  const Kernel& kr = kernel();
  const int MR = kr.MR, NR = kr.NR;
  const int nth = omp_get_max_threads();
  // C tile = MC x NC.  Shrink (never below 2 x 2 register tiles: smaller tiles spend more time packing than
  // multiplying) until every thread has a tile; small problems simply use fewer threads.
  int MC = MR * 8, NC = NR * 16;
  auto tiles = [&](int mc, int nc) { return (long)((M + mc - 1) / mc) * ((N + nc - 1) / nc); };
  while (tiles(MC, NC) < 2L * nth && (MC > 2 * MR || NC > 2 * NR)) {
    if (NC > 2 * NR && (NC / NR >= MC / MR || MC <= 2 * MR)) NC /= 2;
    else MC /= 2;
  }
  const int tm = (M + MC - 1) / MC, tn = (N + NC - 1) / NC;
  const int use = (int)std::min<long>(nth, (long)tm * tn);
#pragma omp parallel num_threads(use)
  {
    // packing buffers live for the life of the thread (a fresh 600 KB vector per thread per call is an mmap + page
    // faults every time: it dominated the step on a 128-core host)
    static thread_local std::vector<float> Ap, Bp;
    if (Ap.size() < (size_t)MC * KC) Ap.resize((size_t)MC * KC);
    if (Bp.size() < (size_t)NC * KC) Bp.resize((size_t)NC * KC);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int im = 0; im < tm; ++im)
      for (int in = 0; in < tn; ++in) {
        const int i0 = im * MC, j0 = in * NC;
        const int mc = std::min(MC, M - i0), nc = std::min(NC, N - j0);
        for (int p0 = 0; p0 < K; p0 += KC) {
          const int kc = std::min(KC, K - p0);
          // pack A: panels of MR rows, [k][MR]
          for (int ir = 0; ir < mc; ir += MR) {
            float* d = Ap.data() + (size_t)ir * kc;
            const int mr = std::min(MR, mc - ir);
            for (int k = 0; k < kc; ++k) {
              const float* s = A + (long)(i0 + ir) * rsa + (long)(p0 + k) * csa;
              for (int i = 0; i < mr; ++i) d[k * MR + i] = s[(long)i * rsa];
              for (int i = mr; i < MR; ++i) d[k * MR + i] = 0.f;
            }
          }
          // pack B: panels of NR columns, [k][NR]
          for (int jr = 0; jr < nc; jr += NR) {
            float* d = Bp.data() + (size_t)jr * kc;
            const int nr = std::min(NR, nc - jr);
            for (int k = 0; k < kc; ++k) {
              const float* s = B + (long)(p0 + k) * rsb + (long)(j0 + jr) * csb;
              for (int j = 0; j < nr; ++j) d[k * NR + j] = s[(long)j * csb];
              for (int j = nr; j < NR; ++j) d[k * NR + j] = 0.f;
            }
          }
          const bool a2 = acc || p0 > 0;
          for (int jr = 0; jr < nc; jr += NR)
            for (int ir = 0; ir < mc; ir += MR)
              kr.fn(kc, Ap.data() + (size_t)ir * kc, Bp.data() + (size_t)jr * kc, C + (long)(i0 + ir) * ldc + j0 + jr, ldc,
                    std::min(MR, mc - ir), std::min(NR, nc - jr), a2);
        }
      }
  }
}

// row-major helpers
inline void gemm_nn(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, bool acc) {
  gemm(M, N, K, A, lda, 1, B, ldb, 1, C, ldc, acc);
}
inline void gemm_nt(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, bool acc) {
  gemm(M, N, K, A, lda, 1, B, 1, ldb, C, ldc, acc);  // B given as [N x K]
}
inline void gemm_tn(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, bool acc) {
  gemm(M, N, K, A, 1, lda, B, ldb, 1, C, ldc, acc);  // A given as [K x M]
}

inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// column sums: out[j] += sum_i X[i][j]
void colsum_acc(const float* X, long M, int N, float* out) {
  const int nth = omp_get_max_threads();
  static std::vector<double> part;
  part.assign((size_t)nth * N, 0.0);
#pragma omp parallel
  {
    double* p = part.data() + (size_t)omp_get_thread_num() * N;
#pragma omp for schedule(static)
    for (long i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) p[j] += X[i * N + j];
  }
  for (int j = 0; j < N; ++j) {
    double s = 0;
    for (int t = 0; t < nth; ++t) s += part[(size_t)t * N + j];
    out[j] += (float)s;
  }
}

// ------------------------------------------------------------------------------------------------------
// nn.SeqLSTM (Element-Research rnn).  W [(D+H) x 4H] = [Wx ; Wh], gate order i,f,o,g.
// x [T x N x D]; tok [T x N] or null: maskZero() -- rows whose step input is the zero vector (token 0) get
// h = c = gates = 0.  One x_t*Wx and one h_{t-1}*Wh GEMM per step, as Torch7 runs it.
// ------------------------------------------------------------------------------------------------------
struct LstmState {
  int T = 0, N = 0, D = 0, H = 0;
  std::vector<float> gates, h, c;  // [T x N x 4H], [T x N x H] x 2
};

void lstm_forward(const float* x, const float* W, const float* b, const int32_t* tok, int T, int N, int D, int H,
                  LstmState& S) {
  S.T = T; S.N = N; S.D = D; S.H = H;
  const long NH = (long)N * H;
  auto pzero = [](std::vector<float>& v, size_t n) {   // zero fill on all cores (steady state: no reallocation)
    if (v.size() != n) v.resize(n);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) v[i] = 0.f;
  };
  pzero(S.gates, (size_t)T * N * 4 * H);
  pzero(S.h, (size_t)T * NH);
  pzero(S.c, (size_t)T * NH);
  const float *Wx = W, *Wh = W + (long)D * 4 * H;
  std::vector<float> a((size_t)N * 4 * H);
  for (int t = 0; t < T; ++t) {
    gemm_nn(N, 4 * H, D, x + (long)t * N * D, D, Wx, 4 * H, a.data(), 4 * H, false);
    if (t > 0) gemm_nn(N, 4 * H, H, S.h.data() + (long)(t - 1) * NH, H, Wh, 4 * H, a.data(), 4 * H, true);
    float* g = S.gates.data() + (long)t * N * 4 * H;
    float* h = S.h.data() + (long)t * NH;
    float* c = S.c.data() + (long)t * NH;
    const float* cp = t ? S.c.data() + (long)(t - 1) * NH : nullptr;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
      if (tok && tok[(long)t * N + n] == 0) continue;  // maskZero: state and gates stay zero
      const float* ar = a.data() + (long)n * 4 * H;
      float* gr = g + (long)n * 4 * H;
      for (int j = 0; j < H; ++j) {
        const float i_ = sigmoidf_(ar[j] + b[j]);
        const float f_ = sigmoidf_(ar[H + j] + b[H + j]);
        const float o_ = sigmoidf_(ar[2 * H + j] + b[2 * H + j]);
        const float g_ = tanhf(ar[3 * H + j] + b[3 * H + j]);
        const float cv = f_ * (cp ? cp[(long)n * H + j] : 0.f) + i_ * g_;
        gr[j] = i_; gr[H + j] = f_; gr[2 * H + j] = o_; gr[3 * H + j] = g_;
        c[(long)n * H + j] = cv;
        h[(long)n * H + j] = o_ * tanhf(cv);
      }
    }
  }
}

// Backward through time.  dh_seq [T x N x H] or null, dh_last [N x H] or null.  Accumulates dW [(D+H) x 4H], db [4H];
// writes dx [T x N x D] if dx != null.  S.gates is overwritten by da (gradient w.r.t. pre-activations).
void lstm_backward(const float* x, const float* W, LstmState& S, const float* dh_seq, const float* dh_last, float* dW,
                   float* db, float* dx) {
  const int T = S.T, N = S.N, D = S.D, H = S.H;
  const long NH = (long)N * H;
  const float *Wx = W, *Wh = W + (long)D * 4 * H;
  float *dWx = dW, *dWh = dW + (long)D * 4 * H;
  std::vector<float> dh((size_t)NH, 0.f), dc((size_t)NH, 0.f);
  for (int t = T - 1; t >= 0; --t) {
    float* g = S.gates.data() + (long)t * N * 4 * H;
    const float* c = S.c.data() + (long)t * NH;
    const float* cp = t ? S.c.data() + (long)(t - 1) * NH : nullptr;
    // dh = dL/dh_t (from above) + da_{t+1} * Wh^T
    if (t < T - 1)
      gemm_nt(N, H, 4 * H, S.gates.data() + (long)(t + 1) * N * 4 * H, 4 * H, Wh, 4 * H, dh.data(), H, false);
    else
      std::fill(dh.begin(), dh.end(), 0.f);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
      float* gr = g + (long)n * 4 * H;
      for (int j = 0; j < H; ++j) {
        const long o = (long)n * H + j;
        float d = dh[o];
        if (dh_seq) d += dh_seq[(long)t * NH + o];
        if (dh_last && t == T - 1) d += dh_last[o];
        const float i_ = gr[j], f_ = gr[H + j], o_ = gr[2 * H + j], g_ = gr[3 * H + j];
        const float tc = tanhf(c[o]);
        const float dcv = dc[o] + d * o_ * (1.f - tc * tc);
        gr[j] = dcv * g_ * i_ * (1.f - i_);
        gr[H + j] = dcv * (cp ? cp[o] : 0.f) * f_ * (1.f - f_);
        gr[2 * H + j] = d * tc * o_ * (1.f - o_);
        gr[3 * H + j] = dcv * i_ * (1.f - g_ * g_);
        dc[o] = dcv * f_;
      }
    }
    // parameter and input gradients of this step (masked rows have gates == 0 => da == 0)
    gemm_tn(D, 4 * H, N, x + (long)t * N * D, D, g, 4 * H, dWx, 4 * H, true);
    if (t > 0) gemm_tn(H, 4 * H, N, S.h.data() + (long)(t - 1) * NH, H, g, 4 * H, dWh, 4 * H, true);
    colsum_acc(g, N, 4 * H, db);
    if (dx) gemm_nt(N, D, 4 * H, g, 4 * H, Wx, 4 * H, dx + (long)t * N * D, D, false);
  }
}

// y = x W^T + b (nn.Linear, weight [out x in]); optional tanh
void linear_forward(const float* x, long M, int in, int out, const float* W, const float* b, float* y, bool tanh_) {
  gemm_nt((int)M, out, in, x, in, W, in, y, out, false);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < M; ++i)
    for (int j = 0; j < out; ++j) {
      const float v = y[i * out + j] + b[j];
      y[i * out + j] = tanh_ ? tanhf(v) : v;
    }
}
// dy is the gradient w.r.t. the PRE-activation output; dx may be null
void linear_backward(const float* x, long M, int in, int out, const float* W, const float* dy, float* dW, float* db,
                     float* dx) {
  gemm_tn(out, in, (int)M, dy, out, x, in, dW, in, true);
  colsum_acc(dy, M, out, db);
  if (dx) gemm_nn((int)M, in, out, dy, out, W, in, dx, in, false);
}

struct Dims {
  int B, R, Tq, Th, To, O, V, E, H, S2, C, K;
};

struct Offsets {  // flat parameter layout = oracle/visdial_oracle.py param_spec('mn-att-ques-im-hist', 'disc')
  long embed, hist1W, hist1b, hist2W, hist2b, ques1W, ques1b, ques2W, ques2b, mn1W, mn1b, mn2W, mn2b, projW, projb, icW,
      icb, qcW, qcb, attW, attb, outW, outb, optW, optb, total;
};
Offsets offsets(const Dims& d) {
  Offsets o;
  long p = 0;
  auto take = [&](long n) {
    const long r = p;
    p += n;
    return r;
  };
  const long H = d.H, E = d.E, K = d.K, C = d.C;
  o.embed = take((long)(d.V + 1) * E);
  o.hist1W = take((E + H) * 4 * H); o.hist1b = take(4 * H);
  o.hist2W = take((H + H) * 4 * H); o.hist2b = take(4 * H);
  o.ques1W = take((E + H) * 4 * H); o.ques1b = take(4 * H);
  o.ques2W = take((H + H) * 4 * H); o.ques2b = take(4 * H);
  o.mn1W = take(H * H); o.mn1b = take(H);
  o.mn2W = take(H * H); o.mn2b = take(H);
  o.projW = take(H * C); o.projb = take(H);
  o.icW = take(K * H); o.icb = take(K);
  o.qcW = take(K * H); o.qcb = take(K);
  o.attW = take(K); o.attb = take(1);
  o.outW = take(H * H); o.outb = take(H);
  o.optW = take((E + H) * 4 * H); o.optb = take(4 * H);
  o.total = p;
  return o;
}

// nn.LookupTableMaskZero (+ nn.Dropout with an explicit keep mask, scale 1/(1-p) = 2): tok [T x N] time-major
void embed_forward(const float* emb, const int32_t* tok, long rows, int E, const uint8_t* keep, float* x) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r) {
    const int t = tok[r];
    float* d = x + r * E;
    if (t == 0) {
      memset(d, 0, sizeof(float) * E);
      continue;
    }
    const float* s = emb + (long)t * E;
    if (keep)
      for (int j = 0; j < E; ++j) d[j] = keep[r * E + j] ? s[j] * 2.f : 0.f;
    else
      memcpy(d, s, sizeof(float) * E);
  }
}
// dense gradWeight accumulate (pads hit row 0, like the reference's table)
void embed_backward(float* demb, const int32_t* tok, long rows, int E, const uint8_t* keep, const float* dx) {
  for (long r = 0; r < rows; ++r) {  // serial: rows of one token collide
    float* d = demb + (long)tok[r] * E;
    const float* s = dx + r * E;
    if (keep)
      for (int j = 0; j < E; ++j) d[j] += keep[r * E + j] ? s[j] * 2.f : 0.f;
    else
      for (int j = 0; j < E; ++j) d[j] += s[j];
  }
}

void transpose_tok(const int32_t* src, int rows, int T, std::vector<int32_t>& dst) {  // [rows x T] -> [T x rows]
  dst.resize((size_t)rows * T);
  for (int r = 0; r < rows; ++r)
    for (int t = 0; t < T; ++t) dst[(size_t)t * rows + r] = src[(size_t)r * T + t];
}

}  // namespace

extern "C" {

struct vdcpu_dims {
  int32_t B, R, Tq, Th, To, O, V, E, H, S2, C, K;
};

int vdcpu_num_threads(void) { return omp_get_max_threads(); }
void vdcpu_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
const char* vdcpu_gemm_kernel(void) { return kernel().name; }
int64_t vdcpu_num_params(const vdcpu_dims* dd) {
  Dims d{dd->B, dd->R, dd->Tq, dd->Th, dd->To, dd->O, dd->V, dd->E, dd->H, dd->S2, dd->C, dd->K};
  return offsets(d).total;
}

// plain C = A[M x K] * B[K x N] entry point (unit test of the GEMM against numpy)
void vdcpu_gemm(int M, int N, int K, const float* A, int64_t rsa, int64_t csa, const float* B, int64_t rsb, int64_t csb,
                float* C, int64_t ldc, int acc) {
  gemm(M, N, K, A, rsa, csa, B, rsb, csb, C, ldc, acc != 0);
}

// One training step, model.lua:66-106 / :249-342 for mn-att-ques-im-hist + disc.
//   ques [N x Tq], hist [N x Th] right-aligned; options [N x O x To] left-aligned; answer_ind [N] 1-based;
//   img [B x S2 x C]; keep-masks (uint8, null = evaluate mode / Dropout off):
//     m_qemb [Tq x N x E], m_hemb [Th x N x E], m_hatt [N x H], m_imgtr [N x S2 x H], m_iqc [N x S2 x K], m_u [N x H]
//   W: flat parameters (layout = offsets()); G: flat gradients (zeroed here, model.lua:68); scores [N x O] or null.
//   do_update: clamp +-5, adam (optim_updates.lua:62-91) with state m, v and step count t (1-based after increment).
// Returns the loss (mean cross-entropy over the N rounds).
double vdcpu_train_step(const vdcpu_dims* dd, float* W, float* G, const int32_t* ques, const int32_t* hist,
                        const float* img, const int32_t* options, const int32_t* answer_ind, const uint8_t* m_qemb,
                        const uint8_t* m_hemb, const uint8_t* m_hatt, const uint8_t* m_imgtr, const uint8_t* m_iqc,
                        const uint8_t* m_u, float* scores_out, int do_update, float lr, float* adam_m, float* adam_v,
                        int t_adam) {
  const Dims d{dd->B, dd->R, dd->Tq, dd->Th, dd->To, dd->O, dd->V, dd->E, dd->H, dd->S2, dd->C, dd->K};
  const Offsets o = offsets(d);
  const int B = d.B, R = d.R, N = B * R, H = d.H, E = d.E, K = d.K, C = d.C, S2 = d.S2, O = d.O, NO = N * d.O;
  memset(G, 0, sizeof(float) * o.total);                      // wrapper:zeroGradParameters()
  memset(W + o.embed, 0, sizeof(float) * E);                  // LookupTableMaskZero zeroes the pad row

  // ---- input re-layout (model.lua:255-294): time-major tokens, 10x image repeat
  std::vector<int32_t> qtok, htok, otok;
  transpose_tok(ques, N, d.Tq, qtok);
  transpose_tok(hist, N, d.Th, htok);
  transpose_tok(options, NO, d.To, otok);
  std::vector<float> img_rep((size_t)N * S2 * C);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) memcpy(img_rep.data() + (size_t)n * S2 * C, img + (size_t)(n / R) * S2 * C, sizeof(float) * S2 * C);

  // ---- encoder forward (mn-att:21-45): shared embedding -> Dropout(0.5) -> 2 x SeqLSTM(maskZero) -> last step
  std::vector<float> xq((size_t)d.Tq * N * E), xh((size_t)d.Th * N * E);
  embed_forward(W + o.embed, qtok.data(), (long)d.Tq * N, E, m_qemb, xq.data());
  embed_forward(W + o.embed, htok.data(), (long)d.Th * N, E, m_hemb, xh.data());
  LstmState h1, h2, q1, q2;
  lstm_forward(xh.data(), W + o.hist1W, W + o.hist1b, htok.data(), d.Th, N, E, H, h1);
  lstm_forward(h1.h.data(), W + o.hist2W, W + o.hist2b, htok.data(), d.Th, N, H, H, h2);
  lstm_forward(xq.data(), W + o.ques1W, W + o.ques1b, qtok.data(), d.Tq, N, E, H, q1);
  lstm_forward(q1.h.data(), W + o.ques2W, W + o.ques2b, qtok.data(), d.Tq, N, H, H, q2);
  const float* h3 = h2.h.data() + (size_t)(d.Th - 1) * N * H;   // nn.Select(1, -1)
  const float* q3 = q2.h.data() + (size_t)(d.Tq - 1) * N * H;

  // ---- memory attention (mn-att:48-62, MaskSoftMax.lua:5-21): per dialog, facts j <= i visible
  std::vector<float> prob((size_t)N * R), hatt((size_t)N * H);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    const int b = n / R, i = n % R;
    float s[64];
    float mx = -INFINITY;
    for (int j = 0; j < R; ++j) {
      float acc = 0.f;
      const float* hv = h3 + (size_t)(b * R + j) * H;
      for (int k = 0; k < H; ++k) acc += q3[(size_t)n * H + k] * hv[k];
      s[j] = (j > i) ? -9999999.f : acc;   // maskedFill(mask, -9999999), mask = 1 where j > i (model.lua:281-288)
      mx = std::max(mx, s[j]);
    }
    float z = 0.f;
    for (int j = 0; j < R; ++j) {
      s[j] = expf(s[j] - mx);
      z += s[j];
    }
    for (int j = 0; j < R; ++j) prob[(size_t)n * R + j] = s[j] / z;
    for (int k = 0; k < H; ++k) {
      float acc = 0.f;
      for (int j = 0; j < R; ++j) acc += prob[(size_t)n * R + j] * h3[(size_t)(b * R + j) * H + k];
      hatt[(size_t)n * H + k] = acc;
    }
  }
  // mn-att:64-65
  std::vector<float> hatt_d(hatt), hattTr((size_t)N * H), s2((size_t)N * H), qh2((size_t)N * H);
  if (m_hatt)
    for (size_t i = 0; i < hatt_d.size(); ++i) hatt_d[i] = m_hatt[i] ? hatt_d[i] * 2.f : 0.f;
  linear_forward(hatt_d.data(), N, H, H, W + o.mn1W, W + o.mn1b, hattTr.data(), true);
  for (size_t i = 0; i < s2.size(); ++i) s2[i] = hattTr[i] + q3[i];
  linear_forward(s2.data(), N, H, H, W + o.mn2W, W + o.mn2b, qh2.data(), true);

  // ---- SAN image attention, one hop (mn-att:68-104), on the replicated [N x 196 x C] tensor
  const long NS = (long)N * S2;
  std::vector<float> pre((size_t)NS * H), img_tr, img_common((size_t)NS * K), qc((size_t)N * K), iqc((size_t)NS * K),
      tq((size_t)NS * K), patt((size_t)N * S2), u1((size_t)N * H), u1_d((size_t)N * H), enc_out((size_t)N * H);
  linear_forward(img_rep.data(), NS, C, H, W + o.projW, W + o.projb, pre.data(), true);        // mn-att:74-78
  img_tr = pre;
  if (m_imgtr) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < NS * H; ++i) img_tr[i] = m_imgtr[i] ? img_tr[i] * 2.f : 0.f;
  }
  linear_forward(img_tr.data(), NS, H, K, W + o.icW, W + o.icb, img_common.data(), false);     // mn-att:83-84
  linear_forward(qh2.data(), N, H, K, W + o.qcW, W + o.qcb, qc.data(), false);                 // mn-att:88
#pragma omp parallel for schedule(static)
  for (long r = 0; r < NS; ++r) {                                                              // mn-att:89-92
    const long n = r / S2;
    for (int k = 0; k < K; ++k) {
      const float t = tanhf(img_common[r * K + k] + qc[n * K + k]);
      tq[r * K + k] = t;
      iqc[r * K + k] = m_iqc ? (m_iqc[r * K + k] ? t * 2.f : 0.f) : t;
    }
  }
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {                                                                // mn-att:93-102
    std::vector<float> sc(S2);
    float mx = -INFINITY;
    for (int s = 0; s < S2; ++s) {
      float acc = W[o.attb];
      const float* v = iqc.data() + ((size_t)n * S2 + s) * K;
      for (int k = 0; k < K; ++k) acc += v[k] * W[o.attW + k];
      sc[s] = acc;
      mx = std::max(mx, acc);
    }
    float z = 0.f;
    for (int s = 0; s < S2; ++s) {
      sc[s] = expf(sc[s] - mx);
      z += sc[s];
    }
    for (int s = 0; s < S2; ++s) patt[(size_t)n * S2 + s] = sc[s] / z;
    for (int k = 0; k < H; ++k) {
      float acc = 0.f;
      for (int s = 0; s < S2; ++s) acc += patt[(size_t)n * S2 + s] * img_tr[((size_t)n * S2 + s) * H + k];
      u1[(size_t)n * H + k] = acc + qh2[(size_t)n * H + k];
    }
  }
  for (size_t i = 0; i < u1.size(); ++i) u1_d[i] = m_u ? (m_u[i] ? u1[i] * 2.f : 0.f) : u1[i];
  linear_forward(u1_d.data(), N, H, H, W + o.outW, W + o.outb, enc_out.data(), true);          // mn-att:106

  // ---- decoder forward (disc.lua:3-32): option embeddings (no dropout) -> SeqLSTM WITHOUT maskZero -> h_T
  std::vector<float> xo((size_t)d.To * NO * E);
  embed_forward(W + o.embed, otok.data(), (long)d.To * NO, E, nullptr, xo.data());
  LstmState opt;
  lstm_forward(xo.data(), W + o.optW, W + o.optb, nullptr, d.To, NO, E, H, opt);
  const float* optH = opt.h.data() + (size_t)(d.To - 1) * NO * H;
  std::vector<float> scores((size_t)N * O), dscores((size_t)N * O);
#pragma omp parallel for schedule(static)
  for (int r = 0; r < NO; ++r) {
    const int n = r / O;
    float acc = 0.f;
    for (int k = 0; k < H; ++k) acc += optH[(size_t)r * H + k] * enc_out[(size_t)n * H + k];
    scores[r] = acc;
  }
  if (scores_out) memcpy(scores_out, scores.data(), sizeof(float) * scores.size());

  // ---- CrossEntropyCriterion forward + backward (model.lua:330-334): mean over the N rounds
  double loss = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* s = scores.data() + (size_t)n * O;
    float mx = s[0];
    for (int j = 1; j < O; ++j) mx = std::max(mx, s[j]);
    double z = 0.0;
    for (int j = 0; j < O; ++j) z += exp((double)s[j] - mx);
    const double lse = mx + log(z);
    const int gt = answer_ind[n] - 1;
    loss += lse - s[gt];
    for (int j = 0; j < O; ++j) dscores[(size_t)n * O + j] = (float)(exp((double)s[j] - lse) / N);
    dscores[(size_t)n * O + gt] -= 1.f / N;
  }
  loss /= N;

  // ---- decoder backward (model.lua:335): nn.MM backward, option LSTM BPTT, shared embedding gradient
  std::vector<float> d_optH((size_t)NO * H), d_enc((size_t)N * H, 0.f);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n)
    for (int j = 0; j < O; ++j) {
      const float g = dscores[(size_t)n * O + j];
      const size_t r = (size_t)n * O + j;
      for (int k = 0; k < H; ++k) {
        d_optH[r * H + k] = g * enc_out[(size_t)n * H + k];
        d_enc[(size_t)n * H + k] += g * optH[r * H + k];
      }
    }
  {
    std::vector<float> dxo((size_t)d.To * NO * E);
    lstm_backward(xo.data(), W + o.optW, opt, nullptr, d_optH.data(), G + o.optW, G + o.optb, dxo.data());
    embed_backward(G + o.embed, otok.data(), (long)d.To * NO, E, nullptr, dxo.data());
  }

  // ---- encoder backward (model.lua:337)
  // out layer + dropout
  std::vector<float> dpre_o((size_t)N * H), du1d((size_t)N * H), du1((size_t)N * H);
  for (size_t i = 0; i < dpre_o.size(); ++i) dpre_o[i] = d_enc[i] * (1.f - enc_out[i] * enc_out[i]);
  linear_backward(u1_d.data(), N, H, H, W + o.outW, dpre_o.data(), G + o.outW, G + o.outb, du1d.data());
  for (size_t i = 0; i < du1.size(); ++i) du1[i] = m_u ? (m_u[i] ? du1d[i] * 2.f : 0.f) : du1d[i];
  // attention: u1 = sum_s p_s img_tr_s + u0
  std::vector<float> dscore((size_t)N * S2), dimg_tr((size_t)NS * H), dz((size_t)NS * K), dqc((size_t)N * K, 0.f);
  std::vector<double> dattW((size_t)K, 0.0);
  double dattb = 0.0;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    std::vector<float> dp(S2);
    float dot = 0.f;
    for (int s = 0; s < S2; ++s) {
      float acc = 0.f;
      for (int k = 0; k < H; ++k) acc += du1[(size_t)n * H + k] * img_tr[((size_t)n * S2 + s) * H + k];
      dp[s] = acc;
      dot += patt[(size_t)n * S2 + s] * acc;
    }
    for (int s = 0; s < S2; ++s) {
      const float p = patt[(size_t)n * S2 + s];
      dscore[(size_t)n * S2 + s] = p * (dp[s] - dot);                 // SoftMax backward
      for (int k = 0; k < H; ++k) dimg_tr[((size_t)n * S2 + s) * H + k] = p * du1[(size_t)n * H + k];
    }
  }
  for (long r = 0; r < NS; ++r) {                                       // Linear(K, 1) gradients (serial, fp64 sums)
    const float g = dscore[r];
    dattb += g;
    for (int k = 0; k < K; ++k) dattW[k] += (double)g * iqc[r * K + k];
  }
  for (int k = 0; k < K; ++k) G[o.attW + k] += (float)dattW[k];
  G[o.attb] += (float)dattb;
#pragma omp parallel for schedule(static)
  for (long r = 0; r < NS; ++r) {                                       // Dropout + Tanh backward
    for (int k = 0; k < K; ++k) {
      float g = dscore[r] * W[o.attW + k];
      if (m_iqc) g = m_iqc[r * K + k] ? g * 2.f : 0.f;
      dz[r * K + k] = g * (1.f - tq[r * K + k] * tq[r * K + k]);
    }
  }
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n)                                           // nn.Replicate backward: sum over regions
    for (int s = 0; s < S2; ++s)
      for (int k = 0; k < K; ++k) dqc[(size_t)n * K + k] += dz[((size_t)n * S2 + s) * K + k];
  {
    std::vector<float> dimg_tr2((size_t)NS * H);
    linear_backward(img_tr.data(), NS, H, K, W + o.icW, dz.data(), G + o.icW, G + o.icb, dimg_tr2.data());
#pragma omp parallel for schedule(static)
    for (long i = 0; i < NS * H; ++i) {
      float g = dimg_tr[i] + dimg_tr2[i];
      if (m_imgtr) g = m_imgtr[i] ? g * 2.f : 0.f;
      dimg_tr[i] = g * (1.f - pre[i] * pre[i]);                          // through Dropout and Tanh of mn-att:74-77
    }
    linear_backward(img_rep.data(), NS, C, H, W + o.projW, dimg_tr.data(), G + o.projW, G + o.projb, nullptr);
  }
  std::vector<float> du0((size_t)N * H);
  linear_backward(qh2.data(), N, H, K, W + o.qcW, dqc.data(), G + o.qcW, G + o.qcb, du0.data());
  for (size_t i = 0; i < du0.size(); ++i) du0[i] += du1[i];            // residual CAddTable (mn-att:102)
  // memory block
  std::vector<float> ds2p((size_t)N * H), ds2((size_t)N * H), dhtp((size_t)N * H), dhatt_d((size_t)N * H), dq3((size_t)N * H),
      dh3((size_t)N * H, 0.f);
  for (size_t i = 0; i < ds2p.size(); ++i) ds2p[i] = du0[i] * (1.f - qh2[i] * qh2[i]);
  linear_backward(s2.data(), N, H, H, W + o.mn2W, ds2p.data(), G + o.mn2W, G + o.mn2b, ds2.data());
  for (size_t i = 0; i < dhtp.size(); ++i) dhtp[i] = ds2[i] * (1.f - hattTr[i] * hattTr[i]);
  linear_backward(hatt_d.data(), N, H, H, W + o.mn1W, dhtp.data(), G + o.mn1W, G + o.mn1b, dhatt_d.data());
  if (m_hatt)
    for (size_t i = 0; i < dhatt_d.size(); ++i) dhatt_d[i] = m_hatt[i] ? dhatt_d[i] * 2.f : 0.f;
  for (int b = 0; b < B; ++b) {                                         // MM / MaskSoftMax / MM backward (MaskSoftMax.lua:23-46)
    for (int i = 0; i < R; ++i) {
      const size_t n = (size_t)b * R + i;
      float dp[64], ds[64];
      float dot = 0.f;
      for (int j = 0; j < R; ++j) {
        float acc = 0.f;
        for (int k = 0; k < H; ++k) acc += dhatt_d[n * H + k] * h3[((size_t)b * R + j) * H + k];
        dp[j] = acc;
        dot += prob[n * R + j] * acc;
      }
      for (int j = 0; j < R; ++j) ds[j] = prob[n * R + j] * (dp[j] - dot);
      for (int k = 0; k < H; ++k) {
        float acc = 0.f;
        for (int j = 0; j < R; ++j) acc += ds[j] * h3[((size_t)b * R + j) * H + k];
        dq3[n * H + k] = ds2[n * H + k] + acc;                           // + the CAddTable path of mn-att:65
      }
      for (int j = 0; j < R; ++j)
        for (int k = 0; k < H; ++k)
          dh3[((size_t)b * R + j) * H + k] += prob[n * R + j] * dhatt_d[n * H + k] + ds[j] * q3[n * H + k];
    }
  }
  // text branches: 2 x SeqLSTM BPTT each, gradient arrives at the last step only
  {
    std::vector<float> dx2((size_t)d.Tq * N * H), dx1((size_t)d.Tq * N * E);
    lstm_backward(q1.h.data(), W + o.ques2W, q2, nullptr, dq3.data(), G + o.ques2W, G + o.ques2b, dx2.data());
    lstm_backward(xq.data(), W + o.ques1W, q1, dx2.data(), nullptr, G + o.ques1W, G + o.ques1b, dx1.data());
    embed_backward(G + o.embed, qtok.data(), (long)d.Tq * N, E, m_qemb, dx1.data());
  }
  {
    std::vector<float> dx2((size_t)d.Th * N * H), dx1((size_t)d.Th * N * E);
    lstm_backward(h1.h.data(), W + o.hist2W, h2, nullptr, dh3.data(), G + o.hist2W, G + o.hist2b, dx2.data());
    lstm_backward(xh.data(), W + o.hist1W, h1, dx2.data(), nullptr, G + o.hist1W, G + o.hist1b, dx1.data());
    embed_backward(G + o.embed, htok.data(), (long)d.Th * N, E, m_hemb, dx1.data());
  }

  // ---- clamp + adam (model.lua:96-99; optim_updates.lua:62-91: eps is added to the UNcorrected sqrt(v))
  if (do_update) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float step = (float)(lr * sqrt(1.0 - pow((double)b2, t_adam)) / (1.0 - pow((double)b1, t_adam)));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < o.total; ++i) {
      float g = std::min(5.f, std::max(-5.f, G[i]));
      G[i] = g;
      adam_m[i] = b1 * adam_m[i] + (1.f - b1) * g;
      adam_v[i] = b2 * adam_v[i] + (1.f - b2) * g * g;
      W[i] -= step * adam_m[i] / (sqrtf(adam_v[i]) + eps);
    }
  }
  return loss;
}

}  // extern "C"
