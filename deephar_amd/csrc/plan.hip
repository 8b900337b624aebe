// C-level plan executor: dh_plan_create / dh_forward / dh_forward_host / dh_plan_destroy (include/deephar_hip.h).
// "a plan/execute pair over device pointers ... one plan per device; stream-ordered, no internal threads".
// The blob is written by deephar_amd/engine/serialize.py (format documented there) from a bound, autotuned plan: the
// launch list of Model.predict with every device pointer expressed relative to the activation arena or the weight
// image.  This file owns the two allocations, patches the pointers once and replays the launches in order on the
// caller's stream -- the same C-ABI entry points the Python executor calls, so the results are the same bits.
#include <cstring>
#include <vector>
#include "dh_kernels.h"

namespace {

using dh::check_launch;

enum Fn { F_CONV, F_DW, F_POOL, F_UPADD, F_ELT, F_SAM, F_CTX, F_DMEANS, F_SAM1D, F_KRON, F_GMM, F_COPY, F_ZPAD, F_DFM, F_SAMCTX, F_NORM, F_GROUP, F_PAIR, F_SEG, F_COUNT };

struct Step { int fn; std::vector<unsigned char> payload; };
struct In { uint64_t tag; size_t items; int u8; char* dst; };   // dst: where dh_forward copies the caller's data (patched)
struct Out { size_t off, npix; int C, ld; };

}  // namespace

struct dh_plan {
  int n = 0;
  size_t arena_bytes = 0, weight_bytes = 0, byte_bytes = 0;
  char* arena = nullptr;
  char* weights = nullptr;
  char* bytes = nullptr;        // region 3: uint8 input staging of a uint8-input plan
  std::vector<In> ins;
  std::vector<Out> outs;
  std::vector<Step> steps;
  // dh_forward_host staging (allocated on first use)
  std::vector<float*> in_dev, out_dev;
  hipStream_t stream = nullptr;
};

namespace {

struct Reader {
  const unsigned char* p;
  const unsigned char* end;
  bool ok = true;
  template <typename T>
  T get() {
    T v{};
    if (p + sizeof(T) > end) { ok = false; return v; }
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
};

// (region << 60) | offset -> device pointer; false when the offset leaves its region
bool fix(const dh_plan& pl, uint64_t v, void** out) {
  const uint64_t region = v >> 60, off = v & ((1ull << 60) - 1);
  if (v == 0) { *out = nullptr; return true; }
  if (region == 1 && off < pl.arena_bytes) { *out = pl.arena + off; return true; }
  if (region == 2 && off < pl.weight_bytes) { *out = pl.weights + off; return true; }
  if (region == 3 && off < pl.byte_bytes) { *out = pl.bytes + off; return true; }
  return false;
}

// leading pointer fields of the argument structs (include/deephar_hip.h)
int struct_pointers(int fn) {
  switch (fn) {
    case F_CONV: return 10;
    case F_DW: return 5;
    case F_POOL: return 2;
    case F_ELT: return 6;
    case F_SAM: return 8;
    case F_SAMCTX: return 8;
  }
  return -1;
}
// u64 arguments that follow the struct, and which of them are pointers
void struct_extra(int fn, int* nargs, unsigned* ptr_mask) {
  *nargs = 0;
  *ptr_mask = 0;
  if (fn == F_CONV) *nargs = 1;                                  // tile_cfg
  if (fn == F_SAMCTX) { *nargs = 5; *ptr_mask = 0x08; }           // J, nctx, agg_alpha, y, ldy
}
size_t struct_size(int fn) {
  switch (fn) {
    case F_CONV: return sizeof(dh_conv_args);
    case F_DW: return sizeof(dh_dw_args);
    case F_POOL: return sizeof(dh_pool_args);
    case F_ELT: return sizeof(dh_elt_args);
    case F_SAM: return sizeof(dh_sam_args);
    case F_SAMCTX: return sizeof(dh_sam_args);
  }
  return 0;
}
// number of u64 arguments of the scalar-argument entry points, and which of them are pointers (bit mask)
bool scalar_sig(int fn, int* nargs, unsigned* ptr_mask) {
  switch (fn) {
    case F_UPADD: *nargs = 10; *ptr_mask = 0x15; return true;         // a, lda, b, ldb, y, ldy, N, H, W, C
    case F_CTX: *nargs = 9; *ptr_mask = 0x0f; return true;            // ys, yc, pc, y, F, J, nctx, alpha, ldy
    case F_DMEANS: *nargs = 8; *ptr_mask = 0x0d; return true;         // h, ldh, hxy, hz, F, HW, D, J
    case F_SAM1D: *nargs = 8; *ptr_mask = 0x17; return true;          // hz, grid, z, ldz, vz, F, D, J
    case F_KRON: *nargs = 10; *ptr_mask = 0x15; return true;          // hm, ldh, x, ldx, f, ldf, B, P, J, C
    case F_GMM: *nargs = 7; *ptr_mask = 0x05; return true;            // x, ldx, y, B, P, C, softmax
    case F_COPY: *nargs = 6; *ptr_mask = 0x05; return true;           // x, ldx, y, ldy, npix, C
    case F_ZPAD: *nargs = 10; *ptr_mask = 0x03; return true;          // x, y, B, H, W, C, OH, OW, PT, PL
    case F_DFM: *nargs = 9; *ptr_mask = 0x15; return true;            // d, ldd, h, ldh, z, ldz, F, HW, J
    case F_NORM: *nargs = 5; *ptr_mask = 0x07; return true;           // x (bytes), lut, y, n_pixels, C
  }
  return false;
}

int run_step(const Step& st, void* stream) {
  const unsigned char* p = st.payload.data();
  auto u = [&](int i) { uint64_t v; std::memcpy(&v, p + 8 * i, 8); return v; };
  auto P = [&](int i) { return reinterpret_cast<float*>(static_cast<uintptr_t>(u(i))); };
  auto I = [&](int i) { return static_cast<int>(static_cast<int64_t>(u(i))); };
  auto Fl = [&](int i) { float f; std::memcpy(&f, p + 8 * i, 4); return f; };
  switch (st.fn) {
    case F_CONV: {
      int64_t cfg;
      std::memcpy(&cfg, p + sizeof(dh_conv_args), 8);
      return dh_conv2d_f32(reinterpret_cast<const dh_conv_args*>(p), (int)cfg, stream);
    }
    case F_SAMCTX: {
      const unsigned char* e = p + sizeof(dh_sam_args);
      int64_t J, nctx, ldy;
      float alpha;
      void* y;
      std::memcpy(&J, e, 8); std::memcpy(&nctx, e + 8, 8); std::memcpy(&alpha, e + 16, 4);
      std::memcpy(&y, e + 24, 8); std::memcpy(&ldy, e + 32, 8);
      return dh_softargmax2d_context_f32(reinterpret_cast<const dh_sam_args*>(p), (int)J, (int)nctx, alpha,
                                         static_cast<float*>(y), (int)ldy, stream);
    }
    case F_DW: return dh_dwconv2d_f32(reinterpret_cast<const dh_dw_args*>(p), stream);
    case F_GROUP:
      return dh_conv2d_dw_group_f32(reinterpret_cast<const dh_conv_args*>(p),
                                    reinterpret_cast<const dh_dw_args*>(p + sizeof(dh_conv_args)), stream);
    case F_PAIR:
      return dh_conv2d_pair_f32(reinterpret_cast<const dh_conv_args*>(p), reinterpret_cast<const dh_conv_args*>(p + sizeof(dh_conv_args)),
                                stream);
    case F_SEG:
      return dh_conv2d_seg_f32(reinterpret_cast<const dh_conv_args*>(p), reinterpret_cast<const dh_conv_seg*>(p + sizeof(dh_conv_args)),
                               stream);
    case F_POOL: return dh_pool2d_f32(reinterpret_cast<const dh_pool_args*>(p), stream);
    case F_ELT: return dh_eltwise_f32(reinterpret_cast<const dh_elt_args*>(p), stream);
    case F_SAM: return dh_softargmax2d_f32(reinterpret_cast<const dh_sam_args*>(p), stream);
    case F_UPADD: return dh_upsample2x_add_f32(P(0), I(1), P(2), I(3), P(4), I(5), I(6), I(7), I(8), I(9), stream);
    case F_CTX: return dh_context_aggregation_f32(P(0), P(1), P(2), P(3), I(4), I(5), I(6), Fl(7), I(8), stream);
    case F_DMEANS: return dh_depth_means_f32(P(0), I(1), P(2), P(3), I(4), I(5), I(6), I(7), stream);
    case F_SAM1D: return dh_softargmax1d_f32(P(0), P(1), P(2), I(3), P(4), I(5), I(6), I(7), stream);
    case F_KRON: return dh_kronecker_f32(P(0), I(1), P(2), I(3), P(4), I(5), I(6), I(7), I(8), I(9), stream);
    case F_GMM: return dh_global_maxmin_softmax_f32(P(0), I(1), P(2), I(3), I(4), I(5), I(6), stream);
    case F_COPY: return dh_copy_channels_f32(P(0), I(1), P(2), I(3), static_cast<int64_t>(u(4)), I(5), stream);
    case F_ZPAD: return dh_zeropad2d_f32(P(0), P(1), I(2), I(3), I(4), I(5), I(6), I(7), I(8), I(9), stream);
    case F_DFM: return dh_depth_from_maps_f32(P(0), I(1), P(2), I(3), P(4), I(5), I(6), I(7), I(8), stream);
    case F_NORM: return dh_normalize_u8_f32(reinterpret_cast<const uint8_t*>(P(0)), P(1), P(2), static_cast<int64_t>(u(3)),
                                            I(4), stream);
  }
  return DH_EINVAL;
}

}  // namespace

extern "C" {

int dh_plan_create(const void* blob, size_t blob_bytes, dh_plan** out) {
  if (blob == nullptr || out == nullptr || blob_bytes < 44) return DH_EINVAL;
  Reader r{static_cast<const unsigned char*>(blob), static_cast<const unsigned char*>(blob) + blob_bytes};
  if (std::memcmp(r.p, "DHPL", 4) != 0) return DH_EINVAL;
  r.p += 4;
  const uint32_t version = r.get<uint32_t>();
  if (version != 1 && version != 2) return DH_EINVAL;
  dh_plan* pl = new dh_plan();
  pl->n = r.get<int32_t>();
  pl->arena_bytes = r.get<uint64_t>();
  pl->weight_bytes = r.get<uint64_t>();
  const uint32_t nin = r.get<uint32_t>(), nout = r.get<uint32_t>(), nsteps = r.get<uint32_t>();
  if (version >= 2) pl->byte_bytes = r.get<uint64_t>();
  auto fail = [&](int rc) { dh_plan_destroy(pl); return rc; };
  if (!r.ok || pl->n <= 0 || nin == 0 || nout == 0 || nin > 64 || nout > 4096 || nsteps > (1u << 20) ||
      pl->byte_bytes > (1ull << 40))
    return fail(DH_EINVAL);
  for (uint32_t i = 0; i < nin; ++i) {
    In in{};
    in.tag = r.get<uint64_t>();
    in.items = (size_t)r.get<uint64_t>();
    if (version >= 2) {
      in.u8 = (int)r.get<uint32_t>();
      (void)r.get<uint32_t>();
    } else {
      in.tag |= 1ull << 60;        // version 1 stored the bare arena offset
    }
    pl->ins.push_back(in);
  }
  for (uint32_t i = 0; i < nout; ++i) {
    Out o;
    o.off = r.get<uint64_t>(); o.npix = r.get<uint64_t>(); o.C = (int)r.get<uint32_t>(); o.ld = (int)r.get<uint32_t>();
    pl->outs.push_back(o);
  }
  for (uint32_t i = 0; i < nsteps && r.ok; ++i) {
    Step st;
    st.fn = (int)r.get<uint32_t>();
    const uint32_t nb = r.get<uint32_t>();
    if (!r.ok || st.fn < 0 || st.fn >= F_COUNT || r.p + nb > r.end) return fail(DH_EINVAL);
    st.payload.assign(r.p, r.p + nb);
    r.p += nb;
    pl->steps.push_back(std::move(st));
  }
  if (!r.ok || (size_t)(r.end - r.p) != pl->weight_bytes) return fail(DH_EINVAL);
  if (hipMalloc(&pl->arena, pl->arena_bytes ? pl->arena_bytes : 16) != hipSuccess) return fail(DH_ELAUNCH);
  if (hipMalloc(&pl->weights, pl->weight_bytes ? pl->weight_bytes : 16) != hipSuccess) return fail(DH_ELAUNCH);
  if (pl->weight_bytes && hipMemcpy(pl->weights, r.p, pl->weight_bytes, hipMemcpyHostToDevice) != hipSuccess)
    return fail(DH_ELAUNCH);
  if (pl->byte_bytes && hipMalloc(&pl->bytes, pl->byte_bytes) != hipSuccess) return fail(DH_ELAUNCH);
  // extents, overflow-safe: every factor is bounded by the arena size (in floats) before it is multiplied.  The blob is
  // trusted, code-equivalent input (it carries launch arguments); these checks catch truncation and mix-ups, they are
  // not a sandbox.
  const uint64_t arena_f = pl->arena_bytes / 4, nn = (uint64_t)pl->n;
  auto fits = [&](uint64_t off_bytes, uint64_t rows, uint64_t pitch, uint64_t last) {   // off + ((rows-1)*pitch + last)*4
    if (off_bytes % 4 || off_bytes / 4 > arena_f || rows == 0 || pitch > arena_f || last > arena_f || rows > arena_f) return false;
    const uint64_t room = arena_f - off_bytes / 4;
    if (pitch != 0 && rows - 1 > room / pitch) return false;
    return (rows - 1) * pitch + last <= room;
  };
  for (In& in : pl->ins) {
    const uint64_t region = in.tag >> 60, off = in.tag & ((1ull << 60) - 1);
    if (in.u8 == 0) {
      if (region != 1 || in.items == 0 || in.items > arena_f || !fits(off, nn, in.items, in.items)) return fail(DH_EINVAL);
      in.dst = pl->arena + off;
    } else {                       // uint8 frames land in region 3
      if (in.u8 != 1 || region != 3 || in.items == 0 || in.items > pl->byte_bytes || nn > pl->byte_bytes / in.items ||
          off > pl->byte_bytes - in.items * nn)
        return fail(DH_EINVAL);
      in.dst = pl->bytes + off;
    }
  }
  for (const Out& o : pl->outs)
    if (o.C <= 0 || o.ld < o.C || o.npix == 0 || o.npix > arena_f || !fits(o.off, o.npix * nn, (uint64_t)o.ld, (uint64_t)o.C))
      return fail(DH_EINVAL);
  // patch the pointers (once)
  for (Step& st : pl->steps) {
    const int np = struct_pointers(st.fn);
    if (st.fn == F_GROUP || st.fn == F_PAIR || st.fn == F_SEG) {   // dh_conv_args followed by dh_dw_args (dh_conv2d_dw_group_f32) / by a
                                                   // second dh_conv_args (dh_conv2d_pair_f32) / by dh_conv_seg (dh_conv2d_seg_f32)
      const size_t second = st.fn == F_GROUP ? sizeof(dh_dw_args) : (st.fn == F_PAIR ? sizeof(dh_conv_args) : sizeof(dh_conv_seg));
      if (st.payload.size() != sizeof(dh_conv_args) + second) return fail(DH_EINVAL);
      size_t at_off[10 + 1 + 10 + 1];
      int k = 0;
      for (int i = 0; i < 10; ++i) at_off[k++] = (size_t)8 * i;
      at_off[k++] = offsetof(dh_conv_args, y_pool);
      for (int i = 0; i < (st.fn == F_GROUP ? 5 : (st.fn == F_PAIR ? 10 : 1)); ++i) at_off[k++] = sizeof(dh_conv_args) + (size_t)8 * i;
      if (st.fn == F_PAIR) at_off[k++] = sizeof(dh_conv_args) + offsetof(dh_conv_args, y_pool);
      for (int i = 0; i < k; ++i) {
        unsigned char* at = st.payload.data() + at_off[i];
        uint64_t v;
        void* ptr;
        std::memcpy(&v, at, 8);
        if (!fix(*pl, v, &ptr)) return fail(DH_EINVAL);
        std::memcpy(at, &ptr, 8);
      }
    } else if (np >= 0) {
      int nextra;
      unsigned emask;
      struct_extra(st.fn, &nextra, &emask);
      if (st.payload.size() != struct_size(st.fn) + (size_t)nextra * 8) return fail(DH_EINVAL);
      if (st.fn == F_CONV) {                     // (y_pool sits behind the integer fields of dh_conv_args)
        unsigned char* at = st.payload.data() + offsetof(dh_conv_args, y_pool);
        uint64_t v;
        void* ptr;
        std::memcpy(&v, at, 8);
        if (!fix(*pl, v, &ptr)) return fail(DH_EINVAL);
        std::memcpy(at, &ptr, 8);
      }
      for (int i = 0; i < np + nextra; ++i) {
        if (i >= np && !(emask & (1u << (i - np)))) continue;
        unsigned char* at = st.payload.data() + (i < np ? (size_t)8 * i : struct_size(st.fn) + (size_t)8 * (i - np));
        uint64_t v;
        void* ptr;
        std::memcpy(&v, at, 8);
        if (!fix(*pl, v, &ptr)) return fail(DH_EINVAL);
        std::memcpy(at, &ptr, 8);
      }
    } else {
      int nargs;
      unsigned mask;
      if (!scalar_sig(st.fn, &nargs, &mask) || st.payload.size() != (size_t)nargs * 8) return fail(DH_EINVAL);
      for (int i = 0; i < nargs; ++i)
        if (mask & (1u << i)) {
          uint64_t v;
          void* ptr;
          std::memcpy(&v, st.payload.data() + 8 * i, 8);
          if (!fix(*pl, v, &ptr)) return fail(DH_EINVAL);
          std::memcpy(st.payload.data() + 8 * i, &ptr, 8);
        }
    }
  }
  *out = pl;
  return DH_OK;
}

int dh_plan_destroy(dh_plan* pl) {
  if (pl == nullptr) return DH_OK;
  if (pl->stream) hipStreamSynchronize(pl->stream);
  for (float* q : pl->in_dev) hipFree(q);
  for (float* q : pl->out_dev) hipFree(q);
  if (pl->stream) hipStreamDestroy(pl->stream);
  if (pl->arena) hipFree(pl->arena);
  if (pl->weights) hipFree(pl->weights);
  if (pl->bytes) hipFree(pl->bytes);
  delete pl;
  return DH_OK;
}

int dh_plan_batch(const dh_plan* pl) { return pl ? pl->n : 0; }
int dh_plan_num_inputs(const dh_plan* pl) { return pl ? (int)pl->ins.size() : 0; }
int dh_plan_num_outputs(const dh_plan* pl) { return pl ? (int)pl->outs.size() : 0; }
int64_t dh_plan_input_items(const dh_plan* pl, int i) {
  return (pl && i >= 0 && i < (int)pl->ins.size()) ? (int64_t)pl->ins[i].items : -1;
}
int dh_plan_input_is_u8(const dh_plan* pl, int i) {
  return (pl && i >= 0 && i < (int)pl->ins.size()) ? pl->ins[i].u8 : -1;
}
int64_t dh_plan_output_items(const dh_plan* pl, int i) {
  return (pl && i >= 0 && i < (int)pl->outs.size()) ? (int64_t)(pl->outs[i].npix * pl->outs[i].C) : -1;
}

int dh_forward(dh_plan* pl, const float* const* inputs, int m, float* const* outputs, void* stream) {
  if (pl == nullptr || inputs == nullptr || outputs == nullptr || m <= 0 || m > pl->n) return DH_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (size_t i = 0; i < pl->ins.size(); ++i) {
    if (inputs[i] == nullptr) return DH_EINVAL;
    const In& in = pl->ins[i];
    if (hipMemcpyAsync(in.dst, inputs[i], in.items * (in.u8 ? 1 : 4) * (size_t)m, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return DH_ELAUNCH;
  }
  for (const Step& st : pl->steps) {
    const int rc = run_step(st, stream);
    if (rc != DH_OK) return rc;
  }
  for (size_t i = 0; i < pl->outs.size(); ++i) {
    const Out& o = pl->outs[i];
    if (outputs[i] == nullptr) continue;                      // an output the caller does not want
    const int rc = dh_copy_channels_f32(reinterpret_cast<const float*>(pl->arena + o.off), o.ld, outputs[i], o.C,
                                        (int64_t)o.npix * m, o.C, stream);
    if (rc != DH_OK) return rc;
  }
  return DH_OK;
}

int dh_forward_host(dh_plan* pl, const float* const* inputs_host, int m, float* const* outputs_host) {
  if (pl == nullptr || inputs_host == nullptr || outputs_host == nullptr || m <= 0 || m > pl->n) return DH_EINVAL;
  if (pl->stream == nullptr) {
    // staging is built in locals and committed only when every allocation succeeded: a failure part-way leaves the plan
    // as it was (the next call tries again) instead of with short in_dev / out_dev vectors
    hipStream_t st = nullptr;
    std::vector<float*> in_dev, out_dev;
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    for (size_t i = 0; ok && i < pl->ins.size(); ++i) {
      float* q = nullptr;
      ok = hipMalloc(&q, pl->ins[i].items * (pl->ins[i].u8 ? 1 : 4) * (size_t)pl->n) == hipSuccess;
      if (ok) in_dev.push_back(q);
    }
    for (size_t i = 0; ok && i < pl->outs.size(); ++i) {
      float* q = nullptr;
      ok = hipMalloc(&q, pl->outs[i].npix * pl->outs[i].C * 4 * (size_t)pl->n) == hipSuccess;
      if (ok) out_dev.push_back(q);
    }
    if (!ok) {
      for (float* q : in_dev) hipFree(q);
      for (float* q : out_dev) hipFree(q);
      if (st) hipStreamDestroy(st);
      return DH_ELAUNCH;
    }
    pl->in_dev.swap(in_dev);
    pl->out_dev.swap(out_dev);
    pl->stream = st;
  }
  for (size_t i = 0; i < pl->ins.size(); ++i)
    if (hipMemcpyAsync(pl->in_dev[i], inputs_host[i], pl->ins[i].items * (pl->ins[i].u8 ? 1 : 4) * (size_t)m,
                       hipMemcpyHostToDevice, pl->stream) != hipSuccess)
      return DH_ELAUNCH;
  const int rc = dh_forward(pl, pl->in_dev.data(), m, pl->out_dev.data(), pl->stream);
  if (rc != DH_OK) return rc;
  for (size_t i = 0; i < pl->outs.size(); ++i)
    if (outputs_host[i] != nullptr &&
        hipMemcpyAsync(outputs_host[i], pl->out_dev[i], pl->outs[i].npix * pl->outs[i].C * 4 * (size_t)m,
                       hipMemcpyDeviceToHost, pl->stream) != hipSuccess)
      return DH_ELAUNCH;
  return hipStreamSynchronize(pl->stream) == hipSuccess ? DH_OK : DH_ELAUNCH;
}

}  // extern "C"
