// vp9hip_decoder.hip — frame-level driver (see include/vp9hip_decoder.h): device frame pool,
// work-list / coefficient transfers, phase sequencing.  C-style host code + two small kernels for
// the residual-plane mode.  Reference citations are relative to /root/reference/.
#include <stdlib.h>

#include "../../include/vp9hip_decoder.h"
#include "vp9hip_internal.h"

namespace {

struct DevVec {
  void *p;
  size_t cap;
};

struct Slot {
  vp9hip_frame f;
  size_t bytes[3];
  int ss;
  bool used;
};

}  // namespace

// One set of work lists: the packer that builds them (its arrays are page-locked host memory), their
// device copies, and two events — `uploaded` (copy stream: lists + coefficients are in HBM) and `done`
// (launch stream: the kernels that read them have finished).  The decoder owns a RING of sets (SURVEY
// §8f-1): frame N+1 is packed and uploaded while the kernels of frame N run.
struct ListSet {
  vp9hip_packer *pk;
  DevVec d_inter, d_txb, d_isl_tasks, d_islands, d_wave_off, d_big_tasks, d_lfm, d_coeffs, d_sb_expected;
  vp9hip_frame_params params;
  vp9hip_packed packed;
  int32_t *big_wave_start;  // host copy (read at launch time)
  size_t big_wave_cap;
  hipEvent_t uploaded, done;
  bool begun, have_coeffs, done_pending;
  bool coeff16;  // the frame's coefficient slots are int16 (vp9hip_coeff_layout.narrow)
};

struct vp9hip_decoder {
  vp9hip_ctx *ctx;
  hipStream_t copy_stream;
  hipStream_t dl_stream;  // frame fetches that must not queue behind the next frame's kernels
  char err[512];
  Slot slots[VP9HIP_POOL_SLOTS];
  ListSet sets[VP9HIP_RING_SETS];
  int cur, next;
  DevVec d_res[3];
  int32_t res_stride[3];
  bool have_res;
  bool timed;
  bool timing_off;  // vp9hip_decoder_set_timing(dec, 0): no event pair around a run
};

#define DEC_FAIL(dec, code, ...)                              \
  do {                                                        \
    snprintf((dec)->err, sizeof((dec)->err), __VA_ARGS__);    \
    return (code);                                            \
  } while (0)

#define DEC_HIP(dec, expr)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      DEC_FAIL(dec, VP9HIP_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// propagate an error of the batched layer
#define DEC_CTX(dec, expr)                                                       \
  do {                                                                           \
    int rc_ = (expr);                                                            \
    if (rc_ != VP9HIP_OK) DEC_FAIL(dec, rc_, "%s", vp9hip_last_error((dec)->ctx)); \
  } while (0)

static const int TIMER_RUN = VP9HIP_TIMER_SLOTS - 1;

static int dv_reserve(vp9hip_decoder *dec, DevVec *v, size_t bytes) {
  if (bytes <= v->cap) return VP9HIP_OK;
  DEC_HIP(dec, hipStreamSynchronize(dec->ctx->stream));
  if (v->p) (void)hipFree(v->p);
  v->p = NULL;
  v->cap = 0;
  size_t want = bytes + (bytes >> 2) + 256;
  DEC_HIP(dec, hipMalloc(&v->p, want));
  v->cap = want;
  return VP9HIP_OK;
}

static int dv_upload_on(vp9hip_decoder *dec, DevVec *v, const void *src, size_t bytes, hipStream_t st) {
  int rc = dv_reserve(dec, v, bytes ? bytes : 16);
  if (rc) return rc;
  if (bytes) DEC_HIP(dec, hipMemcpyAsync(v->p, src, bytes, hipMemcpyHostToDevice, st));
  return VP9HIP_OK;
}

static int dv_upload(vp9hip_decoder *dec, DevVec *v, const void *src, size_t bytes) {
  return dv_upload_on(dec, v, src, bytes, dec->ctx->stream);
}

// ---- a frame's uploads as ONE launch -------------------------------------------------------------------------
// A frame of a stream with eight tile columns is 24 coefficient regions + 8 lists: 32 hipMemcpyAsync calls, ~0.3 ms of
// the submitting thread per frame — as much as packing the frame (vp9hip_dec --stats: "pack + launch" 0.77 ms beside a
// parse of 1.07 ms).  The sources are page-locked and therefore readable from the device: a kernel on the copy stream
// gathers all segments, 8 KB per workgroup, 16 bytes per lane and access.  Pageable sources keep hipMemcpyAsync.
constexpr int UP_MAX_SEG = 56;
constexpr unsigned UP_CHUNK = 8192;
struct UpSeg {
  const char *src;
  char *dst;
  unsigned bytes, wg0;
};
struct UpPlan {
  UpSeg seg[UP_MAX_SEG];
  int n;
  unsigned wgs;
};

__global__ __launch_bounds__(256) void gather_upload_kernel(UpPlan pl) {
  const unsigned b = blockIdx.x;
  int s = 0;
  while (s + 1 < pl.n && b >= pl.seg[s + 1].wg0) ++s;
  const UpSeg g = pl.seg[s];
  const unsigned off = (b - g.wg0) * UP_CHUNK;
  const unsigned n = g.bytes - off < UP_CHUNK ? g.bytes - off : UP_CHUNK;
  const char *src = g.src + off;
  char *dst = g.dst + off;
  unsigned done = 0;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    for (unsigned i = threadIdx.x * 16; i + 16 <= n; i += 256 * 16) *(uint4 *)(dst + i) = *(const uint4 *)(src + i);
    done = n & ~15u;
  }
  if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
    for (unsigned i = done + threadIdx.x * 4; i + 4 <= n; i += 256 * 4) *(unsigned *)(dst + i) = *(const unsigned *)(src + i);
    done += (n - done) & ~3u;
  }
  for (unsigned i = done + threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// the device's address of page-locked host memory (hipHostMalloc: the same address; registered memory: its mapping),
// NULL for pageable memory
static const char *host_device_address(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (a pageable pointer is reported as an error: clear it)
    return NULL;
  }
  return a.type == hipMemoryTypeHost ? (const char *)a.devicePointer : NULL;
}

static int up_flush(vp9hip_decoder *dec, UpPlan *pl, hipStream_t st) {
  if (pl->n == 0) return VP9HIP_OK;
  hipLaunchKernelGGL(gather_upload_kernel, dim3(pl->wgs), dim3(256), 0, st, *pl);
  DEC_HIP(dec, hipGetLastError());
  pl->n = 0;
  pl->wgs = 0;
  return VP9HIP_OK;
}

// one more segment of the frame's uploads: src (page-locked if `pinned`) -> dst
static int up_add(vp9hip_decoder *dec, UpPlan *pl, void *dst, const void *src, size_t bytes, bool pinned, hipStream_t st) {
  if (bytes == 0) return VP9HIP_OK;
  if (!pinned || bytes > 0xffffffffu - UP_CHUNK) {
    DEC_HIP(dec, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    return VP9HIP_OK;
  }
  if (pl->n == UP_MAX_SEG) {
    int rc = up_flush(dec, pl, st);
    if (rc) return rc;
  }
  UpSeg *g = &pl->seg[pl->n++];
  g->src = (const char *)src;
  g->dst = (char *)dst;
  g->bytes = (unsigned)bytes;
  g->wg0 = pl->wgs;
  pl->wgs += (unsigned)((bytes + UP_CHUNK - 1) / UP_CHUNK);
  return VP9HIP_OK;
}

static int up_vec(vp9hip_decoder *dec, UpPlan *pl, DevVec *v, const void *src, size_t bytes, hipStream_t st) {
  int rc = dv_reserve(dec, v, bytes ? bytes : 16);
  if (rc) return rc;
  return up_add(dec, pl, v->p, src, bytes, true /* the packer's vectors: pinned_alloc */, st);
}

static void *pinned_alloc(void *user, size_t bytes) {
  (void)user;
  void *p = NULL;
  return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : NULL;
}
static void pinned_free(void *user, void *p) {
  (void)user;
  (void)hipHostFree(p);
}

extern "C" int vp9hip_decoder_create(int device, vp9hip_decoder **out) {
  if (!out) return VP9HIP_EINVAL;
  *out = NULL;
  vp9hip_decoder *dec = (vp9hip_decoder *)calloc(1, sizeof(*dec));
  if (!dec) return VP9HIP_ENOMEM;
  int rc = vp9hip_create(device, &dec->ctx);
  if (rc) {
    free(dec);
    return rc;  // text in vp9hip_last_error(NULL)
  }
  bool ok = hipStreamCreateWithFlags(&dec->copy_stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < VP9HIP_RING_SETS && ok; ++i) {
    ListSet *s = &dec->sets[i];
    ok = vp9hip_packer_create_ex(&s->pk, pinned_alloc, pinned_free, NULL) == VP9HIP_OK &&
         hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s->done, hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) {
    vp9hip_decoder_destroy(dec);
    return VP9HIP_ENOMEM;
  }
  *out = dec;
  return VP9HIP_OK;
}

extern "C" void vp9hip_decoder_destroy(vp9hip_decoder *dec) {
  if (!dec) return;
  (void)hipSetDevice(dec->ctx->device);
  (void)hipStreamSynchronize(dec->ctx->stream);
  if (dec->copy_stream) (void)hipStreamSynchronize(dec->copy_stream);
  if (dec->dl_stream) (void)hipStreamSynchronize(dec->dl_stream);
  for (int s = 0; s < VP9HIP_POOL_SLOTS; ++s)
    for (int p = 0; p < 3; ++p)
      if (dec->slots[s].f.plane[p]) (void)hipFree(dec->slots[s].f.plane[p]);
  for (int i = 0; i < VP9HIP_RING_SETS; ++i) {
    ListSet *s = &dec->sets[i];
    DevVec *all[] = { &s->d_inter, &s->d_txb, &s->d_isl_tasks, &s->d_islands, &s->d_wave_off, &s->d_big_tasks,
                      &s->d_lfm,   &s->d_coeffs, &s->d_sb_expected };
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); ++k)
      if (all[k]->p) (void)hipFree(all[k]->p);
    free(s->big_wave_start);
    if (s->pk) vp9hip_packer_destroy(s->pk);
    if (s->uploaded) (void)hipEventDestroy(s->uploaded);
    if (s->done) (void)hipEventDestroy(s->done);
  }
  for (int p = 0; p < 3; ++p)
    if (dec->d_res[p].p) (void)hipFree(dec->d_res[p].p);
  if (dec->copy_stream) (void)hipStreamDestroy(dec->copy_stream);
  if (dec->dl_stream) (void)hipStreamDestroy(dec->dl_stream);
  vp9hip_destroy(dec->ctx);
  free(dec);
}

extern "C" void *vp9hip_decoder_host_alloc(vp9hip_decoder *dec, size_t bytes) {
  if (!dec || hipSetDevice(dec->ctx->device) != hipSuccess) return NULL;
  return pinned_alloc(NULL, bytes);
}

extern "C" void vp9hip_decoder_host_free(vp9hip_decoder *dec, void *p) {
  if (dec && p) pinned_free(NULL, p);
}

extern "C" const char *vp9hip_decoder_error(const vp9hip_decoder *dec) {
  return dec ? dec->err : vp9hip_last_error(NULL);
}

extern "C" vp9hip_ctx *vp9hip_decoder_ctx(vp9hip_decoder *dec) { return dec ? dec->ctx : NULL; }

extern "C" int vp9hip_decoder_alloc_slot(vp9hip_decoder *dec, int slot, int width, int height, int ss, int bit_depth,
                                         int hbd, int clear) {
  if (!dec) return VP9HIP_EINVAL;
  if (slot < 0 || slot >= VP9HIP_POOL_SLOTS) DEC_FAIL(dec, VP9HIP_EINVAL, "pool slot %d out of range", slot);
  if (width <= 0 || height <= 0 || width > 16384 || height > 16384 || (ss != 0 && ss != 1) ||
      (bit_depth != 8 && bit_depth != 10 && bit_depth != 12) || (bit_depth > 8 && !hbd))
    DEC_FAIL(dec, VP9HIP_EINVAL, "bad frame geometry %dx%d ss %d bd %d hbd %d", width, height, ss, bit_depth, hbd);
  DEC_HIP(dec, hipSetDevice(dec->ctx->device));
  Slot *s = &dec->slots[slot];
  const int aw = (width + 7) & ~7, ah = (height + 7) & ~7;
  const int bps = hbd ? 2 : 1;
  vp9hip_frame f;
  memset(&f, 0, sizeof(f));
  f.bit_depth = bit_depth;
  f.hbd = hbd ? 1 : 0;
  for (int p = 0; p < 3; ++p) {
    const int sh = p ? ss : 0;
    f.width[p] = (width + sh) >> sh;
    f.height[p] = (height + sh) >> sh;
    f.awidth[p] = aw >> sh;
    f.aheight[p] = ah >> sh;
    f.stride[p] = (f.awidth[p] + 63) & ~63;
  }
  bool same = s->used && s->f.bit_depth == f.bit_depth && s->f.hbd == f.hbd && s->ss == ss;
  for (int p = 0; p < 3 && same; ++p)
    same = s->f.width[p] == f.width[p] && s->f.height[p] == f.height[p] && s->f.stride[p] == f.stride[p];
  if (!same) {
    DEC_HIP(dec, hipStreamSynchronize(dec->ctx->stream));
    for (int p = 0; p < 3; ++p) {
      const size_t need = (size_t)f.stride[p] * f.aheight[p] * bps;
      if (need > s->bytes[p]) {
        if (s->f.plane[p]) (void)hipFree(s->f.plane[p]);
        s->f.plane[p] = NULL;
        s->bytes[p] = 0;
        DEC_HIP(dec, hipMalloc(&s->f.plane[p], need));
        s->bytes[p] = need;
      }
      f.plane[p] = s->f.plane[p];
    }
    s->f = f;
    s->ss = ss;
    s->used = true;
  }
  if (clear)
    for (int p = 0; p < 3; ++p)
      DEC_HIP(dec, hipMemsetAsync(s->f.plane[p], 0, (size_t)s->f.stride[p] * s->f.aheight[p] * bps, dec->ctx->stream));
  return VP9HIP_OK;
}

static int host_frame_ok(const vp9hip_host_frame *h) {
  return h && h->plane[0] && h->plane[1] && h->plane[2] && h->ss_x == h->ss_y;
}

extern "C" int vp9hip_decoder_upload(vp9hip_decoder *dec, int slot, const vp9hip_host_frame *src) {
  if (!dec) return VP9HIP_EINVAL;
  if (!host_frame_ok(src)) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_upload: bad host frame");
  int rc = vp9hip_decoder_alloc_slot(dec, slot, src->width, src->height, src->ss_x, src->bit_depth, src->hbd, 0);
  if (rc) return rc;
  const Slot *s = &dec->slots[slot];
  const int bps = src->hbd ? 2 : 1;
  for (int p = 0; p < 3; ++p) {
    if (src->stride[p] < s->f.awidth[p]) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_upload: plane %d stride too small", p);
    DEC_HIP(dec, hipMemcpy2DAsync(s->f.plane[p], (size_t)s->f.stride[p] * bps, src->plane[p], (size_t)src->stride[p] * bps,
                                  (size_t)s->f.awidth[p] * bps, s->f.aheight[p], hipMemcpyHostToDevice, dec->ctx->stream));
  }
  DEC_HIP(dec, hipStreamSynchronize(dec->ctx->stream));
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_download(vp9hip_decoder *dec, int slot, const vp9hip_host_frame *dst) {
  if (!dec) return VP9HIP_EINVAL;
  if (slot < 0 || slot >= VP9HIP_POOL_SLOTS || !dec->slots[slot].used)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download: slot %d holds no frame", slot);
  if (!host_frame_ok(dst)) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download: bad host frame");
  const Slot *s = &dec->slots[slot];
  if (dst->width != s->f.width[0] || dst->height != s->f.height[0] || (dst->hbd != 0) != (s->f.hbd != 0) ||
      dst->ss_x != s->ss)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download: host frame geometry differs from slot %d", slot);
  const int bps = s->f.hbd ? 2 : 1;
  DEC_HIP(dec, hipSetDevice(dec->ctx->device));
  for (int p = 0; p < 3; ++p) {
    if (dst->stride[p] < s->f.awidth[p]) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download: plane %d stride too small", p);
    DEC_HIP(dec, hipMemcpy2DAsync(dst->plane[p], (size_t)dst->stride[p] * bps, s->f.plane[p], (size_t)s->f.stride[p] * bps,
                                  (size_t)s->f.awidth[p] * bps, s->f.aheight[p], hipMemcpyDeviceToHost, dec->ctx->stream));
  }
  int rc = vp9hip_sync(dec->ctx);
  if (rc) DEC_FAIL(dec, rc, "%s", vp9hip_last_error(dec->ctx));
  return VP9HIP_OK;
}

// A frame fetched as soon as ITS run is through, while later frames' kernels are already queued on the launch
// stream: the copy waits for the `done` event of the ring set the frame was run from, on a stream of its own.
extern "C" int vp9hip_decoder_download_after(vp9hip_decoder *dec, int slot, const vp9hip_host_frame *dst, int ring_set) {
  if (!dec) return VP9HIP_EINVAL;
  if (slot < 0 || slot >= VP9HIP_POOL_SLOTS || !dec->slots[slot].used)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download_after: slot %d holds no frame", slot);
  if (ring_set < 0 || ring_set >= VP9HIP_RING_SETS || !dec->sets[ring_set].done_pending)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download_after: ring set %d has no run in flight", ring_set);
  if (!host_frame_ok(dst)) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download_after: bad host frame");
  const Slot *s = &dec->slots[slot];
  if (dst->width != s->f.width[0] || dst->height != s->f.height[0] || (dst->hbd != 0) != (s->f.hbd != 0) || dst->ss_x != s->ss)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download_after: host frame geometry differs from slot %d", slot);
  const int bps = s->f.hbd ? 2 : 1;
  DEC_HIP(dec, hipSetDevice(dec->ctx->device));
  // created on first use: a process that runs several decoders side by side without ever fetching early keeps the
  // stream-to-hardware-queue layout it had (HIP spreads a process's streams over a handful of hardware queues; one
  // more stream per decoder put all eight launch streams of bench.py's multi-stream leg on the same queue)
  if (!dec->dl_stream) DEC_HIP(dec, hipStreamCreateWithFlags(&dec->dl_stream, hipStreamNonBlocking));
  DEC_HIP(dec, hipStreamWaitEvent(dec->dl_stream, dec->sets[ring_set].done, 0));
  for (int p = 0; p < 3; ++p) {
    if (dst->stride[p] < s->f.awidth[p]) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_download_after: plane %d stride too small", p);
    DEC_HIP(dec, hipMemcpy2DAsync(dst->plane[p], (size_t)dst->stride[p] * bps, s->f.plane[p], (size_t)s->f.stride[p] * bps,
                                  (size_t)s->f.awidth[p] * bps, s->f.aheight[p], hipMemcpyDeviceToHost, dec->dl_stream));
  }
  // a frame whose loop filter gave up waiting is never delivered: the flag is read on the download stream, behind the
  // run's `done` event like the planes (vp9hip_sync reports and clears it at the caller's next synchronisation)
  int lf_err = 0;
  if (dec->ctx->lf_err_flag)
    DEC_HIP(dec, hipMemcpyAsync(&lf_err, dec->ctx->lf_err_flag, sizeof(int), hipMemcpyDeviceToHost, dec->dl_stream));
  DEC_HIP(dec, hipStreamSynchronize(dec->dl_stream));
  if (lf_err)
    DEC_FAIL(dec, VP9HIP_EDEVICE, "vp9hip_decoder_download_after: the frame's loop filter gave up waiting (flag %d); the frame is not valid",
             lf_err);
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_slot_frame(vp9hip_decoder *dec, int slot, vp9hip_frame *out) {
  if (!dec || !out) return VP9HIP_EINVAL;
  if (slot < 0 || slot >= VP9HIP_POOL_SLOTS || !dec->slots[slot].used)
    DEC_FAIL(dec, VP9HIP_EINVAL, "slot %d holds no frame", slot);
  *out = dec->slots[slot].f;
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_begin_frame_ex(vp9hip_decoder *dec, const vp9hip_frame_params *params,
                                             const vp9hip_block *blocks, int n_blocks, const vp9hip_coeff_layout *layout,
                                             const int32_t *const dqcoeff[3], int flags) {
  if (!dec) return VP9HIP_EINVAL;
  if (!params) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: null params");
  if (dqcoeff && !layout) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: dqcoeff without the eob layout");
  DEC_HIP(dec, hipSetDevice(dec->ctx->device));
  ListSet *S = &dec->sets[dec->next];
  dec->cur = dec->next;
  dec->next = (dec->next + 1) % VP9HIP_RING_SETS;
  S->begun = false;
  dec->have_res = false;
  // this set's previous frame: its kernels must be through with the device lists, and (earlier on the
  // timeline) its copies with the packer's page-locked arrays, before either is overwritten
  if (S->done_pending) {
    DEC_HIP(dec, hipEventSynchronize(S->done));
    S->done_pending = false;
  } else {
    DEC_HIP(dec, hipEventSynchronize(S->uploaded));  // begun but never run
  }
  const hipStream_t cs = dec->copy_stream;
  int rc;
  // slots placed by the caller (block_off): where they go on the device does not depend on the packer, so the
  // coefficients — the bulk of a frame's bytes — start travelling before the lists are built
  const bool early = dqcoeff && layout->block_off;
  // int16 slots (vp9hip_coeff_layout.narrow): same offsets in coefficients, half the bytes on the way to the device
  const bool narrow = dqcoeff && layout && layout->narrow != 0;
  if (narrow && !layout->block_off) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: int16 slots need caller-placed slots (block_off)");
  const size_t esz = narrow ? sizeof(int16_t) : sizeof(int32_t);
  UpPlan up;
  up.n = 0;
  up.wgs = 0;
  const char *coef_dev[3] = { NULL, NULL, NULL };  // the coefficient arrays as the device sees them (page-locked ones)
  if (dqcoeff && (flags & VP9HIP_BEGIN_HOST_PERSISTENT))
    for (int p = 0; p < 3; ++p) coef_dev[p] = dqcoeff[p] ? host_device_address(dqcoeff[p]) : NULL;
  if (early) {
    if (layout->total < 0 || layout->total > (int64_t)UINT32_MAX)
      DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: bad coefficient total");
    if ((rc = dv_reserve(dec, &S->d_coeffs, esz * (size_t)(layout->total + 16)))) return rc;
    for (int64_t r = 0; r < layout->n_regions; ++r) {
      const vp9hip_coeff_region *g = &layout->regions[r];
      if (g->plane < 0 || g->plane > 2 || g->start < 0 || g->count < 0 ||
          layout->plane_base[g->plane] < 0 || layout->plane_base[g->plane] + g->start + g->count > layout->total || !dqcoeff[g->plane]) {
        (void)hipStreamSynchronize(cs);
        DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: coefficient region %lld out of range", (long long)r);
      }
      if (g->count &&
          (rc = up_add(dec, &up, (char *)S->d_coeffs.p + esz * (size_t)(layout->plane_base[g->plane] + g->start),
                       (coef_dev[g->plane] ? coef_dev[g->plane] : (const char *)dqcoeff[g->plane]) + esz * (size_t)g->start,
                       esz * (size_t)g->count, coef_dev[g->plane] != NULL, cs)))
        return rc;
    }
    if ((rc = up_flush(dec, &up, cs))) return rc;  // (the coefficients travel while the lists are built)
  }
  rc = vp9hip_pack_frame(S->pk, params, blocks, n_blocks, layout, &S->packed);
  if (rc) {
    if (early) (void)hipStreamSynchronize(cs);  // the caller's arrays are its own again
    DEC_FAIL(dec, rc, "%s", vp9hip_packer_error(S->pk));
  }
  const vp9hip_packed *P = &S->packed;
  S->params = *params;
  if ((rc = up_vec(dec, &up, &S->d_inter, P->inter, sizeof(vp9hip_inter_task) * (size_t)P->n_inter, cs))) return rc;
  if ((rc = up_vec(dec, &up, &S->d_txb, P->txb, sizeof(vp9hip_txb) * (size_t)P->n_txb, cs))) return rc;
  if ((rc = up_vec(dec, &up, &S->d_isl_tasks, P->intra_island_tasks, sizeof(vp9hip_intra_task) * (size_t)P->n_intra_island_tasks, cs)))
    return rc;
  if ((rc = up_vec(dec, &up, &S->d_islands, P->islands, sizeof(vp9hip_intra_island) * (size_t)P->n_islands, cs))) return rc;
  if ((rc = up_vec(dec, &up, &S->d_wave_off, P->island_wave_off, sizeof(int32_t) * (size_t)P->n_island_wave_off, cs))) return rc;
  if ((rc = up_vec(dec, &up, &S->d_big_tasks, P->intra_big_tasks, sizeof(vp9hip_intra_task) * (size_t)P->n_intra_big_tasks, cs)))
    return rc;
  if (P->lfm && (rc = up_vec(dec, &up, &S->d_lfm, P->lfm, sizeof(vp9hip_lfm) * (size_t)P->sb_rows * P->sb_cols, cs))) return rc;
  if ((size_t)(P->n_big_waves + 1) > S->big_wave_cap) {
    free(S->big_wave_start);
    S->big_wave_cap = (size_t)P->n_big_waves + 64;
    S->big_wave_start = (int32_t *)malloc(sizeof(int32_t) * S->big_wave_cap);
    if (!S->big_wave_start) {
      S->big_wave_cap = 0;
      DEC_FAIL(dec, VP9HIP_ENOMEM, "out of host memory");
    }
  }
  memcpy(S->big_wave_start, P->big_wave_start, sizeof(int32_t) * (size_t)(P->n_big_waves + 1));
  if (P->island_sb_expected &&
      (rc = up_vec(dec, &up, &S->d_sb_expected, P->island_sb_expected, sizeof(int32_t) * (size_t)P->sb_rows * P->sb_cols, cs)))
    return rc;
  if (!early && (rc = dv_reserve(dec, &S->d_coeffs, esz * (size_t)(P->coeff_total + 16)))) return rc;
  S->have_coeffs = dqcoeff != NULL;
  S->coeff16 = narrow;
  if (early) {
    if (P->coeff_total != layout->total) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: coefficient total mismatch");
  } else if (dqcoeff && layout->block_off) {
    for (int64_t r = 0; r < layout->n_regions; ++r) {
      const vp9hip_coeff_region *g = &layout->regions[r];
      if (g->plane < 0 || g->plane > 2 || g->start < 0 || g->count < 0 ||
          layout->plane_base[g->plane] + g->start + g->count > P->coeff_total || !dqcoeff[g->plane])
        DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: coefficient region %lld out of range", (long long)r);
      if (g->count &&
          (rc = up_add(dec, &up, (char *)S->d_coeffs.p + esz * (size_t)(layout->plane_base[g->plane] + g->start),
                       (coef_dev[g->plane] ? coef_dev[g->plane] : (const char *)dqcoeff[g->plane]) + esz * (size_t)g->start,
                       esz * (size_t)g->count, coef_dev[g->plane] != NULL, cs)))
        return rc;
    }
  } else if (dqcoeff)
    for (int p = 0; p < 3; ++p)
      if (P->coeff_count[p]) {
        if (!dqcoeff[p]) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_begin_frame: dqcoeff[%d] is null", p);
        DEC_HIP(dec, hipMemcpyAsync((int32_t *)S->d_coeffs.p + P->coeff_base[p], dqcoeff[p],
                                    sizeof(int32_t) * (size_t)P->coeff_count[p], hipMemcpyHostToDevice, cs));
      }
  if ((rc = up_flush(dec, &up, cs))) return rc;
  DEC_HIP(dec, hipEventRecord(S->uploaded, cs));
  // a caller that may reuse its coefficient arrays right away (the reference frees them at the end of
  // decode_tiles) gets the synchronous contract; page-locked arrays that live until the frame has been
  // run (VP9HIP_BEGIN_HOST_PERSISTENT) travel while the caller goes on
  if (dqcoeff && !(flags & VP9HIP_BEGIN_HOST_PERSISTENT)) DEC_HIP(dec, hipEventSynchronize(S->uploaded));
  S->begun = true;
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_begin_frame(vp9hip_decoder *dec, const vp9hip_frame_params *params,
                                          const vp9hip_block *blocks, int n_blocks, const vp9hip_coeff_layout *layout,
                                          const int32_t *const dqcoeff[3]) {
  return vp9hip_decoder_begin_frame_ex(dec, params, blocks, n_blocks, layout, dqcoeff, 0);
}

extern "C" int vp9hip_decoder_current_set(const vp9hip_decoder *dec) { return dec ? dec->cur : -1; }

extern "C" int vp9hip_decoder_select_set(vp9hip_decoder *dec, int set) {
  if (!dec) return VP9HIP_EINVAL;
  if (set < 0 || set >= VP9HIP_RING_SETS || !dec->sets[set].begun)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_select_set: set %d holds no frame", set);
  dec->cur = set;
  dec->have_res = false;
  return VP9HIP_OK;
}

// ---- residual-plane mode ----------------------------------------------------------------------
namespace {

// dst = clip(dst + residual) over a whole plane: the reference's block_sum / inter_residual_sum
// (libvpx/vp9/decoder/vp9_decodeframe.c:290-341, 1117-1148; vpx-master/inter_cuda_kernel.cu:821-829)
// for every sample at once — the residual plane is zero where nothing was coded.
__global__ __launch_bounds__(256) void residual_add_plane_kernel(uint16_t *__restrict__ dst, int dstride,
                                                                 const int64_t *__restrict__ res, int rstride, int w, int h,
                                                                 int maxv) {
  const int x = (blockIdx.x * 256 + threadIdx.x) * 2, y = blockIdx.y;
  if (x >= w || y >= h) return;
  // two samples per lane: one 32-bit pixel access, one 16-byte residual access (w is a multiple of 4)
  const longlong2 r = *(const longlong2 *)(res + (size_t)y * rstride + x);
  uint32_t *dp = (uint32_t *)(dst + (size_t)y * dstride + x);
  const uint32_t px = *dp;
  int a = (int)(px & 0xffff) + (int)r.x, b = (int)(px >> 16) + (int)r.y;
  a = a < 0 ? 0 : a > maxv ? maxv : a;
  b = b < 0 ? 0 : b > maxv ? maxv : b;
  *dp = (uint32_t)a | ((uint32_t)b << 16);
}

// Gather the residual of every coded intra transform block into its coefficient slot (int32,
// raster N x N) so that the intra kernel can add it right after predicting the block
// (tx_type bit 6).  One workgroup per task.
__global__ __launch_bounds__(256) void residual_gather_kernel(const vp9hip_intra_task *__restrict__ tasks, int n,
                                                              const int64_t *r0, const int64_t *r1, const int64_t *r2,
                                                              int s0, int s1, int s2, int h0, int h12,
                                                              int32_t *__restrict__ coeffs) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const vp9hip_intra_task tk = tasks[i];
  if (tk.eob == 0) return;
  const int bs = 4 << tk.tx_size;
  const int64_t *res = tk.plane == 0 ? r0 : tk.plane == 1 ? r1 : r2;
  const int rs = tk.plane == 0 ? s0 : tk.plane == 1 ? s1 : s2;  // = plane width on the device
  const int ph = tk.plane == 0 ? h0 : h12;
  for (int k = threadIdx.x; k < bs * bs; k += 256) {
    const int r = k / bs, c = k % bs;
    // a block may overhang the aligned plane; only its visible part is stored (and later written)
    const bool in = tk.y + r < ph && tk.x + c < rs;
    coeffs[tk.coeff_off + k] = in ? (int32_t)res[(size_t)(tk.y + r) * rs + tk.x + c] : 0;
  }
}

__global__ void mark_identity_kernel(vp9hip_intra_task *tasks, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) tasks[i].tx_type |= 0x40;
}

}  // namespace

extern "C" int vp9hip_decoder_set_residual_planes(vp9hip_decoder *dec, const int64_t *const res[3],
                                                  const int32_t stride[3]) {
  if (!dec) return VP9HIP_EINVAL;
  ListSet *S = &dec->sets[dec->cur];
  if (!S->begun) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_set_residual_planes: no frame begun");
  if (!res || !stride || !res[0] || !res[1] || !res[2]) DEC_FAIL(dec, VP9HIP_EINVAL, "null residual plane");
  if (!S->params.hbd)
    DEC_FAIL(dec, VP9HIP_EINVAL,
             "residual-plane mode needs a high-bitdepth frame (the reference's CPU transforms only produce int64 "
             "residuals on their highbd path); pass the coefficient buffers to vp9hip_decoder_begin_frame instead");
  if (S->have_coeffs) DEC_FAIL(dec, VP9HIP_EINVAL, "coefficients were already given for this frame");
  const int aw = (S->params.width + 7) & ~7, ah = (S->params.height + 7) & ~7, ss = S->params.ss_x;
  DEC_HIP(dec, hipSetDevice(dec->ctx->device));
  for (int p = 0; p < 3; ++p) {
    const int w = p ? aw >> ss : aw, h = p ? ah >> ss : ah;
    if (stride[p] < w) DEC_FAIL(dec, VP9HIP_EINVAL, "residual plane %d stride too small", p);
    int rc = dv_reserve(dec, &dec->d_res[p], sizeof(int64_t) * (size_t)w * h);
    if (rc) return rc;
    dec->res_stride[p] = w;
    DEC_HIP(dec, hipMemcpy2DAsync(dec->d_res[p].p, sizeof(int64_t) * (size_t)w, res[p], sizeof(int64_t) * (size_t)stride[p],
                                  sizeof(int64_t) * (size_t)w, h, hipMemcpyHostToDevice, dec->ctx->stream));
  }
  DEC_HIP(dec, hipStreamSynchronize(dec->ctx->stream));
  dec->have_res = true;
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_run(vp9hip_decoder *dec, int phases, const int ref_slot[3], int dst_slot,
                                  const vp9hip_lfm *h_lfm, const vp9hip_lf_thresh *thresh) {
  if (!dec) return VP9HIP_EINVAL;
  ListSet *S = &dec->sets[dec->cur];
  if (!S->begun) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: no frame begun");
  if (dst_slot < 0 || dst_slot >= VP9HIP_POOL_SLOTS || !dec->slots[dst_slot].used)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: destination slot %d holds no frame", dst_slot);
  const vp9hip_packed *P = &S->packed;
  const vp9hip_frame *dst = &dec->slots[dst_slot].f;
  if (dst->width[0] != S->params.width || dst->height[0] != S->params.height || dst->hbd != (S->params.hbd ? 1 : 0) ||
      dst->bit_depth != S->params.bit_depth)
    DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: destination slot geometry differs from the frame parameters");
  DEC_HIP(dec, hipSetDevice(dec->ctx->device));
  hipStream_t st = dec->ctx->stream;
  // lists + coefficients: usually long in HBM when a replayed or pipelined frame is run — then no
  // dependency packet goes into the queue (every marker / barrier packet costs the command processor
  // microseconds between two kernels)
  if (hipEventQuery(S->uploaded) != hipSuccess) DEC_HIP(dec, hipStreamWaitEvent(st, S->uploaded, 0));
  DEC_CTX(dec, vp9hip_set_coeff_bits(dec->ctx, (S->coeff16 && !dec->have_res) ? 16 : 32));
  if (!dec->timing_off) DEC_CTX(dec, vp9hip_timer_begin(dec->ctx, TIMER_RUN));

  const bool do_pred = (phases & (VP9HIP_PHASE_INTER | VP9HIP_PHASE_INTER_PRED)) != 0;
  const bool do_resid = (phases & (VP9HIP_PHASE_INTER | VP9HIP_PHASE_INTER_RESID)) != 0;
  if (do_pred || do_resid) {
    if (do_pred && P->n_inter) {
      vp9hip_frame refs[3];
      memset(refs, 0, sizeof(refs));
      for (int k = 0; k < 3; ++k) {
        const bool needed = (P->refs_used >> k) & 1;
        const int s = ref_slot ? ref_slot[k] : -1;
        if (s >= 0 && s < VP9HIP_POOL_SLOTS && dec->slots[s].used) {
          refs[k] = dec->slots[s].f;
          if (needed && (refs[k].width[0] != S->params.ref_width[k] || refs[k].height[0] != S->params.ref_height[k]))
            DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: reference %d is %dx%d in the pool, %dx%d in the parameters", k,
                     refs[k].width[0], refs[k].height[0], S->params.ref_width[k], S->params.ref_height[k]);
          if (needed && (refs[k].hbd != dst->hbd || refs[k].bit_depth != dst->bit_depth))
            DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: reference %d has another sample format", k);
        } else if (needed) {
          DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: reference %d is used by the frame but slot %d holds no frame", k, s);
        } else {
          refs[k] = *dst;  // placeholder, never read
        }
      }
      DEC_CTX(dec, vp9hip_inter_pred_batch(dec->ctx, (const vp9hip_inter_task *)S->d_inter.p, P->inter_class_count, refs, 3, dst));
    }
    if (!do_resid) {
    } else if (dec->have_res) {
      for (int p = 0; p < 3; ++p) {
        const int w = dst->awidth[p], h = dst->aheight[p];
        hipLaunchKernelGGL(residual_add_plane_kernel, dim3((w / 2 + 255) / 256, h), dim3(256), 0, st, (uint16_t *)dst->plane[p],
                           dst->stride[p], (const int64_t *)dec->d_res[p].p, dec->res_stride[p], w, h, (1 << dst->bit_depth) - 1);
      }
      DEC_HIP(dec, hipGetLastError());
    } else if (P->n_txb) {
      if (!S->have_coeffs) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: the frame has residuals but no coefficients were given");
      DEC_CTX(dec, vp9hip_idct_add_batch(dec->ctx, (const vp9hip_txb *)S->d_txb.p, P->txb_size_count, (const int32_t *)S->d_coeffs.p, dst));
    }
  }

  if ((phases & VP9HIP_PHASE_INTRA) && P->n_intra) {
    const int32_t *coeffs = (S->have_coeffs || dec->have_res) ? (const int32_t *)S->d_coeffs.p : NULL;
    if (dec->have_res) {
      // residual of coded intra blocks: gather from the planes into the coefficient slots
      vp9hip_intra_task *lists[2] = { (vp9hip_intra_task *)S->d_isl_tasks.p, (vp9hip_intra_task *)S->d_big_tasks.p };
      const int counts[2] = { P->n_intra_island_tasks, P->n_intra_big_tasks };
      for (int l = 0; l < 2; ++l) {
        if (!counts[l]) continue;
        hipLaunchKernelGGL(residual_gather_kernel, dim3(counts[l]), dim3(256), 0, st, lists[l], counts[l],
                           (const int64_t *)dec->d_res[0].p, (const int64_t *)dec->d_res[1].p, (const int64_t *)dec->d_res[2].p,
                           dec->res_stride[0], dec->res_stride[1], dec->res_stride[2], dst->aheight[0], dst->aheight[1],
                           (int32_t *)S->d_coeffs.p);
        hipLaunchKernelGGL(mark_identity_kernel, dim3((counts[l] + 255) / 256), dim3(256), 0, st, lists[l], counts[l]);
      }
      DEC_HIP(dec, hipGetLastError());
    }
    // islands and the loop filter as one launch when both phases are asked for and nothing forces the
    // sequence (key frames' large components; an explicit mask array that must be uploaded first is fine)
    const bool overlap = (phases & VP9HIP_PHASE_LF) && thresh && P->n_islands && !P->n_intra_big_tasks && P->sb_rows <= 128 &&
                         P->sb_cols <= 128 && P->island_sb_expected && P->island_row_pos && (h_lfm || P->lfm);
    if (overlap) {
      if (h_lfm) {
        int rc = dv_upload(dec, &S->d_lfm, h_lfm, sizeof(vp9hip_lfm) * (size_t)P->sb_rows * P->sb_cols);
        if (rc) return rc;
        DEC_HIP(dec, hipStreamSynchronize(st));
      }
      DEC_CTX(dec, vp9hip_intra_islands_lf(dec->ctx, (const vp9hip_intra_task *)S->d_isl_tasks.p,
                                           (const vp9hip_intra_island *)S->d_islands.p, P->n_islands,
                                           (const int32_t *)S->d_wave_off.p, coeffs, (const int32_t *)S->d_sb_expected.p,
                                           P->island_row_pos, (const vp9hip_lfm *)S->d_lfm.p, P->sb_rows, P->sb_cols, thresh, dst, 3));
      phases &= ~VP9HIP_PHASE_LF;
    } else {
      if (P->n_islands)
        DEC_CTX(dec, vp9hip_intra_pred_islands(dec->ctx, (const vp9hip_intra_task *)S->d_isl_tasks.p,
                                               (const vp9hip_intra_island *)S->d_islands.p, P->n_islands,
                                               (const int32_t *)S->d_wave_off.p, coeffs, dst));
      if (P->n_intra_big_tasks)
        DEC_CTX(dec, vp9hip_intra_pred_waves(dec->ctx, (const vp9hip_intra_task *)S->d_big_tasks.p, S->big_wave_start,
                                             P->n_big_waves, coeffs, dst));
    }
  }

  if (phases & VP9HIP_PHASE_LF) {
    if (!thresh) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: loop filter needs the threshold table");
    if (h_lfm) {
      int rc = dv_upload(dec, &S->d_lfm, h_lfm, sizeof(vp9hip_lfm) * (size_t)P->sb_rows * P->sb_cols);
      if (rc) return rc;
      DEC_HIP(dec, hipStreamSynchronize(st));  // h_lfm may be pageable and change after we return
    } else if (!P->lfm) {
      DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_run: no loop-filter masks (set params.build_lf_masks or pass them)");
    }
    DEC_CTX(dec, vp9hip_loop_filter_frame(dec->ctx, (const vp9hip_lfm *)S->d_lfm.p, P->sb_rows, P->sb_cols, thresh, dst,
                                          3));
  }
  if (!dec->timing_off) DEC_CTX(dec, vp9hip_timer_end(dec->ctx, TIMER_RUN));
  DEC_HIP(dec, hipEventRecord(S->done, st));
  S->done_pending = true;
  dec->timed = !dec->timing_off;
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_set_timing(vp9hip_decoder *dec, int on) {
  if (!dec) return VP9HIP_EINVAL;
  dec->timing_off = !on;
  if (!on) dec->timed = false;
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_sync(vp9hip_decoder *dec) {
  if (!dec) return VP9HIP_EINVAL;
  DEC_CTX(dec, vp9hip_sync(dec->ctx));
  return VP9HIP_OK;
}

extern "C" int vp9hip_decoder_last_run_ms(vp9hip_decoder *dec, float *ms) {
  if (!dec || !ms) return VP9HIP_EINVAL;
  if (!dec->timed) DEC_FAIL(dec, VP9HIP_EINVAL, "vp9hip_decoder_last_run_ms: nothing was run");
  DEC_CTX(dec, vp9hip_timer_read(dec->ctx, TIMER_RUN, ms));
  return VP9HIP_OK;
}

extern "C" const vp9hip_packed *vp9hip_decoder_packed(const vp9hip_decoder *dec) {
  return dec && dec->sets[dec->cur].begun ? &dec->sets[dec->cur].packed : NULL;
}
