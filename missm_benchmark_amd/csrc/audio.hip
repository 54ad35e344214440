// Audio front end of the LanguageBind audio model on the GPU (reference languagebind/audio/processing_audio.py:31-111):
//   resample (torchaudio.functional.resample, :44-46)  ->  waveform - mean (:96)  ->  Kaldi filter bank (torchaudio.compliance.kaldi.fbank
//   with htk_compat, hanning window, 25 ms / 10 ms frames, log mel energies, :97-107)  ->  three target_length-frame chunks (or the
//   clip tiled up to target_length), transposed to [3, mel_bins, target_length] and normalised (x - mean) / (2 std) (:54-93).
// torchaudio is a host library that is not in this image and is unpinned upstream; the arithmetic is restated from its published
// algorithm (the CPU twin is oracle/missm_oracle.py: kaldi_fbank / sinc_resample - PARITY UNPINNED, said there too).
//
// One workgroup per frame: the frame sits in LDS (<= 2048 samples after padding to a power of two), DC removal / pre-emphasis / window
// in place, an in-LDS radix-2 FFT (log2(N) barrier-separated stages, exact twiddles from sincospif), power spectrum, then thread b sums
// its triangular mel filter over the FFT bins in bin order (deterministic, no atomics).  HBM traffic is the waveform once (2.5x: frames
// overlap) and the [frames, mel_bins] output: launch-bound at clip lengths of seconds.
#include "common.h"
#include "missm_internal.h"
#include <math.h>

namespace missm {

constexpr int FB_MAXN = 2048;   // padded window (power of two): 25 ms at <= 48 kHz is 1200 samples -> 2048

// mean of a device buffer, one workgroup, fixed summation order (thread-strided partials, then a tree)
__global__ __launch_bounds__(1024) void buffer_mean_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[1024];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}

__device__ __forceinline__ float mel_of(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

struct FbankArgs {
  const float* wave; long n; const float* gmean;   // samples of channel 0; *gmean is subtracted first (may be null)
  float* out;                                      // [frames, num_mel]
  int frames, win, shift, npad, log2n, num_mel;
  float sample_rate, low_freq, high_freq, preemph, eps;
};

__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs a) {
  __shared__ float re[FB_MAXN], im[FB_MAXN];
  __shared__ float red[256];
  __shared__ float melk[FB_MAXN / 2];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int N = a.npad, W = a.win;
  const float gm = a.gmean ? a.gmean[0] : 0.f;
  // ---- frame (snip_edges: frame f starts at f * shift), zero padded to N
  float s = 0.f;
  for (int t = tid; t < N; t += 256) {
    const float v = t < W ? a.wave[(long)f * a.shift + t] - gm : 0.f;
    re[t] = v;
    s += v;
  }
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float dc = red[0] / (float)W;              // remove_dc_offset: the frame's own mean
  __syncthreads();
  // ---- pre-emphasis x[t] - c x[max(t - 1, 0)] on the DC-free frame, then the hanning window 0.5 - 0.5 cos(2 pi t / (W - 1))
  float tmp[FB_MAXN / 256];
#pragma unroll
  for (int q = 0; q < FB_MAXN / 256; ++q) {
    const int t = tid + 256 * q;
    float v = 0.f;
    if (t < W) {
      const float x0 = re[t] - dc, x1 = re[t > 0 ? t - 1 : 0] - dc;
      const float w = 0.5f - 0.5f * cospif(2.0f * (float)t / (float)(W - 1));
      v = (x0 - a.preemph * x1) * w;
    }
    tmp[q] = v;
  }
  __syncthreads();
  // ---- bit-reversed placement, then log2(N) radix-2 stages in LDS
#pragma unroll
  for (int q = 0; q < FB_MAXN / 256; ++q) {
    const int t = tid + 256 * q;
    if (t < N) {
      const int r = (int)(__brev((unsigned)t) >> (32 - a.log2n));
      re[r] = tmp[q];
      im[r] = 0.f;
    }
  }
  __syncthreads();
  for (int st = 1; st <= a.log2n; ++st) {
    const int half = 1 << (st - 1);
    for (int b = tid; b < N / 2; b += 256) {
      const int j = b & (half - 1), base = ((b >> (st - 1)) << st) + j;
      float sn, cs;
      sincospif(-(float)j / (float)half, &sn, &cs);          // exp(-2 pi i j / 2^st)
      const float xr = re[base + half], xi = im[base + half];
      const float tr = xr * cs - xi * sn, ti = xr * sn + xi * cs;
      const float ur = re[base], ui = im[base];
      re[base] = ur + tr; im[base] = ui + ti;
      re[base + half] = ur - tr; im[base + half] = ui - ti;
    }
    __syncthreads();
  }
  // ---- power spectrum of bins 0 .. N/2 - 1 (the Nyquist bin carries a zero filter weight, kaldi pads the bank with a zero column)
  const int nb = N / 2;
  const float binw = a.sample_rate / (float)N;
  for (int k = tid; k < nb; k += 256) {
    const float p = re[k] * re[k] + im[k] * im[k];
    melk[k] = mel_of(binw * (float)k);
    re[k] = p;
  }
  __syncthreads();
  // ---- triangular mel filters: bin b spans [low + b d, low + (b + 2) d] in mel, weight = max(0, min(up, down))
  const float nyq = 0.5f * a.sample_rate;
  const float hi = a.high_freq > 0.f ? a.high_freq : a.high_freq + nyq;
  const float mlo = mel_of(a.low_freq), mhi = mel_of(hi);
  const float d = (mhi - mlo) / (float)(a.num_mel + 1);
  for (int b = tid; b < a.num_mel; b += 256) {
    const float left = mlo + (float)b * d, center = mlo + ((float)b + 1.0f) * d, right = mlo + ((float)b + 2.0f) * d;
    float acc = 0.f;
    for (int k = 0; k < nb; ++k) {
      const float m = melk[k];
      const float up = (m - left) / (center - left), down = (right - m) / (right - center);
      const float w = fmaxf(0.f, fminf(up, down));
      acc += w * re[k];
    }
    a.out[(long)f * a.num_mel + b] = logf(fmaxf(acc, a.eps));
  }
}

// out[c, b, t] = (mel[src(c, t), b] - mean) / (2 std): frames start[c] + t of a clip longer than `target`, frame t mod frames of a
// shorter one (mel.repeat(n)[:target]); the transpose to [mel_bins, target_length] rides along
__global__ __launch_bounds__(256) void mel_assemble_kernel(const float* __restrict__ mel, int frames, int num_mel, float* __restrict__ out,
                                                          int target, int s0, int s1, int s2, float mean, float inv2std) {
  const long total = 3L * num_mel * target;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % target;
    const int b = (idx / target) % num_mel;
    const int c = idx / ((long)target * num_mel);
    const int start = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const int src = frames > target ? start + t : t % frames;
    out[idx] = (mel[(long)src * num_mel + b] - mean) * inv2std;
  }
}

// out[q * new_f + p] = sum_j kern[p, j] * padded[q * orig_f + j], padded = `width` zeros | wave | zeros (torchaudio's strided conv1d)
__global__ __launch_bounds__(256) void sinc_resample_kernel(const float* __restrict__ wave, long n, const float* __restrict__ kern, int klen,
                                                           int orig_f, int new_f, int width, float* __restrict__ out, long n_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const long q = i / new_f;
  const int p = i % new_f;
  const float* kr = kern + (long)p * klen;
  const long base = q * orig_f - width;
  float acc = 0.f;
  for (int j = 0; j < klen; ++j) {
    const long s = base + j;
    if (s >= 0 && s < n) acc += kr[j] * wave[s];
  }
  out[i] = acc;
}

}  // namespace missm

using namespace missm;

extern "C" int missm_buffer_mean(const float* x, long n, float* out, void* stream) {
  MISSM_CHECK_ARG(x && out && n > 0, "buffer_mean: empty buffer");
  hipLaunchKernelGGL(buffer_mean_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), x, n, out);
  return missm_check_launch("buffer_mean");
}

extern "C" int missm_fbank_frames(long n, float sample_rate, float frame_length_ms, float frame_shift_ms) {
  const int win = (int)(sample_rate * 0.001f * frame_length_ms), shift = (int)(sample_rate * 0.001f * frame_shift_ms);
  if (win <= 0 || shift <= 0 || n < win) return 0;
  return 1 + (int)((n - win) / shift);
}

extern "C" int missm_kaldi_fbank(const float* wave, long n, const float* global_mean, float* out, int num_mel_bins, float sample_rate,
                                 float frame_length_ms, float frame_shift_ms, float low_freq, float high_freq, float preemphasis,
                                 void* stream) {
  MISSM_CHECK_ARG(wave && out && n > 0 && num_mel_bins > 0 && num_mel_bins <= 1024 && sample_rate > 0, "kaldi_fbank: bad arguments");
  FbankArgs a;
  a.win = (int)(sample_rate * 0.001f * frame_length_ms);
  a.shift = (int)(sample_rate * 0.001f * frame_shift_ms);
  MISSM_CHECK_ARG(a.win >= 2 && a.shift >= 1, "kaldi_fbank: frame length / shift too small for the sample rate");
  a.npad = 1; a.log2n = 0;
  while (a.npad < a.win) { a.npad <<= 1; ++a.log2n; }           // round_to_power_of_two
  MISSM_CHECK_ARG(a.npad <= FB_MAXN && a.npad >= 2, "kaldi_fbank: window longer than 2048 samples");
  a.frames = missm_fbank_frames(n, sample_rate, frame_length_ms, frame_shift_ms);
  MISSM_CHECK_ARG(a.frames > 0, "kaldi_fbank: waveform shorter than one frame");
  a.wave = wave; a.n = n; a.gmean = global_mean; a.out = out; a.num_mel = num_mel_bins;
  a.sample_rate = sample_rate; a.low_freq = low_freq; a.high_freq = high_freq; a.preemph = preemphasis;
  a.eps = 1.1920928955078125e-07f;                              // torch.finfo(torch.float32).eps
  hipLaunchKernelGGL(fbank_kernel, dim3(a.frames), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return missm_check_launch("kaldi_fbank");
}

extern "C" int missm_mel_assemble(const float* mel, int frames, int num_mel_bins, float* out, int target_length, int start0, int start1,
                                  int start2, float mean, float std, void* stream) {
  MISSM_CHECK_ARG(mel && out && frames > 0 && num_mel_bins > 0 && target_length > 0 && std != 0.f, "mel_assemble: bad arguments");
  const int lim = frames > target_length ? frames - target_length : 0;
  MISSM_CHECK_ARG(start0 >= 0 && start1 >= 0 && start2 >= 0 && start0 <= lim && start1 <= lim && start2 <= lim,
                  "mel_assemble: chunk start outside the clip");
  const long total = 3L * num_mel_bins * target_length;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(mel_assemble_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), mel, frames, num_mel_bins, out,
                     target_length, start0, start1, start2, mean, 1.0f / (2.0f * std));
  return missm_check_launch("mel_assemble");
}

extern "C" int missm_sinc_resample(const float* wave, long n, const float* kernels, int kernel_len, int orig_freq, int new_freq, int width,
                                   float* out, long n_out, void* stream) {
  MISSM_CHECK_ARG(wave && kernels && out && n > 0 && n_out > 0 && kernel_len == 2 * width + orig_freq && orig_freq > 0 && new_freq > 0,
                  "sinc_resample: bad arguments");
  MISSM_CHECK_ARG(n_out <= ((n + (long)orig_freq - 1) / orig_freq + 1) * (long)new_freq, "sinc_resample: output longer than the resampled clip");
  hipLaunchKernelGGL(sinc_resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), wave, n, kernels,
                     kernel_len, orig_freq, new_freq, width, out, n_out);
  return missm_check_launch("sinc_resample");
}
