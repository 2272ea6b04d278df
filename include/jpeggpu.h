/*
 * jpeggpu.h -- C ABI of the MI355X (gfx950) JPEG pixel back end.
 *
 * Drop-in boundary for ONE path of beldaphilippe/jpeg-decoder: the per-MCU pixel pipeline
 *     iquant + izz  ->  idct_opt (Loeffler 8x8)  ->  chroma upsample + ycc2rgb  ->  raster bytes
 * Everything above it (header parse, Huffman VLD, progressive refinement) stays host C code and
 * only has to fill full-frame blocl16_t coefficient planes.
 *
 * All citations are file:line under the reference tree (/root/reference).  Plain pointers and
 * sizes only; no HIP or torch types appear in any signature (a HIP stream is passed as void*).
 *
 * Every entry point runs on the GPU.  There is no CPU fallback: without a usable device the calls
 * fail with JPEGGPU_ERR_NO_DEVICE / JPEGGPU_ERR_HIP and jpeggpu_last_error() says why.
 */
#ifndef JPEGGPU_H
#define JPEGGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* Layout contract (kept verbatim from the reference)                                          */
/* ------------------------------------------------------------------------------------------ */

/* blocl16_t, include/jpeg2ppm.h:31-34 : one block, 64 x int16 in ZIGZAG order, 128 bytes.     */
typedef struct jpeggpu_blocl16_s { int16_t data[64]; } jpeggpu_blocl16_t;
/* bloct16_t, include/jpeg2ppm.h:37-40 : 8x8 int16, stored data[col][row] by izz (src/iqzz.c:23). */
typedef struct jpeggpu_bloct16_s { int16_t data[8][8]; } jpeggpu_bloct16_t;
/* bloctu8_t, include/jpeg2ppm.h:25-28 : 8x8 uint8, stored data[col][row] (src/idct_opt.c:108-115). */
typedef struct jpeggpu_bloctu8_s { uint8_t data[8][8]; } jpeggpu_bloctu8_t;
/* qtable_t, include/img.h:8-11 : 64 x uint16 in ZIGZAG order (as read from DQT).              */
typedef struct jpeggpu_qtable_s { uint16_t data[64]; } jpeggpu_qtable_t;
/* rgb_t, include/ycc2rgb.h:7-10.                                                              */
typedef struct jpeggpu_rgb_s { uint8_t r, g, b; } jpeggpu_rgb_t;

/* Status codes.  0 and JPEGGPU_ERR_NB_COMP keep the numeric values of erreur_code_t
 * (include/erreur.h:9,31): ERR_NB_COMP is the only error the reference's L1b can raise
 * (src/decoder_utils.c:50,67).  Negative values are this library's own.                       */
enum {
   JPEGGPU_SUCCESS          = 0,
   JPEGGPU_ERR_NB_COMP      = 13,
   JPEGGPU_ERR_INVALID_ARG  = -1,
   JPEGGPU_ERR_NO_DEVICE    = -2,
   JPEGGPU_ERR_HIP          = -3,
   JPEGGPU_ERR_SAMPLING     = -4,   /* max_h % hs != 0 or max_v % vs != 0: the reference's integer
                                       division (src/decoder_utils.c:102-107) indexes out of bounds */
   JPEGGPU_ERR_NO_QT        = -5    /* qtables[idq] == NULL (src/progressive.c:262-267)            */
};

/* One frame component: idcomp_t fields the back end needs (include/img.h:36-43).              */
typedef struct jpeggpu_component_s {
   uint8_t hs;     /* hsampling */
   uint8_t vs;     /* vsampling */
   uint8_t idq;    /* quantisation table index, 0..3 */
   uint8_t _pad;
} jpeggpu_component_t;

/* What L2 has in hand at src/progressive.c:257 / src/baseline.c:160 (img_t, include/img.h:88-103).
 * Components are in FRAME order and must be Y, Cb, Cr (the only order for which the reference's
 * get_ycc_info, src/decoder_utils.c:92-110, is coherent).                                      */
typedef struct jpeggpu_frame_desc_s {
   uint16_t width;          /* img_t.width  (include/img.h:90) */
   uint16_t height;         /* img_t.height (include/img.h:89); only the header uses it: the raster
                               holds nbmcuV*max_v*8 rows (src/decoder_utils.c:51,68)              */
   uint8_t  nbcomp;         /* 1 (gray, P5) or 3 (YCbCr -> RGB, P6); anything else: ERR_NB_COMP    */
   uint8_t  max_h, max_v;   /* img_t.max_hsampling / max_vsampling (include/img.h:97-98)           */
   uint8_t  apply_iquant;   /* 1: planes hold QUANTISED coefficients, multiply by the table
                               (progressive, src/progressive.c:277-279);
                               0: planes are already dequantised (baseline ran iquant inside
                               decode_bloc_baseline, src/baseline.c:66)                            */
   uint32_t nbmcuH, nbmcuV; /* img_t.nbmcuH / nbmcuV (include/img.h:99-100)                        */
   jpeggpu_component_t comp[3];
} jpeggpu_frame_desc_t;

/* ------------------------------------------------------------------------------------------ */
/* Library state                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* Creates the DEFAULT context on HIP device <device>: the stream, staging buffers and "-f" option that the
 * process-wide entry points below (jpeggpu_backend_run ...) use.  Idempotent; a second device is refused
 * here -- further devices, or several independent callers on one device, take contexts of their own
 * (jpeggpu_ctx_create).  The first process-wide call without a prior jpeggpu_init() binds device 0.
 * No counterpart in the reference (one thread, no device).                                          */
int  jpeggpu_init(int device);
/* Destroys the default context.  Not to be called while another thread is inside the library.      */
void jpeggpu_shutdown(void);

/* Per-device context ("our handle is per-device, one host thread per GPU, streams inside").
 * The reference keeps its state in one global, all_option (src/options.c:13), and runs one thread
 * (src/jpeg2ppm.c:62-138); a C caller that drives several GPUs from one process, or several threads on
 * one GPU, gives each its own context: own stream, own staging buffers, own idct_fast option, own
 * "-f" coefficient table.
 *
 * Threading contract.  HIP's current device is a per-thread setting: EVERY entry point of this library
 * binds the calling thread to its context's device first (hipSetDevice), so any thread may call in.
 * A context serialises its host-pointer entries internally (they share its staging buffers); the
 * device-pointer entries only enqueue work and may be called concurrently.  Destroying a context
 * while another thread uses it is the caller's bug.                                                 */
typedef struct jpeggpu_ctx_s jpeggpu_ctx_t;
int  jpeggpu_ctx_create(int device, jpeggpu_ctx_t **ctx);
void jpeggpu_ctx_destroy(jpeggpu_ctx_t *ctx);
int  jpeggpu_ctx_device(const jpeggpu_ctx_t *ctx);
/* all_option.idct_fast of this context (src/options.c:76-79): starts as the process-wide value.     */
void jpeggpu_ctx_set_idct_fast(jpeggpu_ctx_t *ctx, int fast);
int  jpeggpu_ctx_get_idct_fast(const jpeggpu_ctx_t *ctx);
/* Text of the last error raised on the calling thread ("" if none).  Stands in for erreur_t.com
 * (include/erreur.h:81-86).                                                                    */
const char *jpeggpu_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* Geometry helpers (host only, no GPU needed)                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Fills max_h, max_v, nbmcuH, nbmcuV from width, height, nbcomp, comp[].hs/vs exactly like
 * calcul_image_information (src/entete.c:176-197).                                             */
int    jpeggpu_frame_geometry(jpeggpu_frame_desc_t *desc);
/* Blocks in component c's plane: (nbmcuH*hs_c) * (nbmcuV*vs_c)  (src/progressive.c:76-78).     */
size_t jpeggpu_plane_blocks(const jpeggpu_frame_desc_t *desc, int c);
/* Raster rows actually written: nbmcuV*max_v*8 (>= height; src/decoder_utils.c:51,68).         */
size_t jpeggpu_raster_rows(const jpeggpu_frame_desc_t *desc);
/* Bytes per raster row: width * (3 if nbcomp==3 else 1) (src/decoder_utils.c:58,87).           */
size_t jpeggpu_raster_row_bytes(const jpeggpu_frame_desc_t *desc);

/* ------------------------------------------------------------------------------------------ */
/* Frame-level back end: replaces the block loop + save_mcu_ligne_* of                         */
/* src/progressive.c:257-312 and src/baseline.c:160-217                                        */
/* ------------------------------------------------------------------------------------------ */

/* coef[c] : plane of component c, blocl16_t[(nbmcuV*vs_c) * (nbmcuH*hs_c)], block (blocX,blocY)
 *           at index blocY*nbH_c + blocX (src/progressive.c:118,157,274-278), contiguous.
 * qtables : up to 4 tables indexed by comp[c].idq; may hold NULLs for unused ids; ignored when
 *           apply_iquant == 0.
 * out     : raster, jpeggpu_raster_rows() rows of jpeggpu_raster_row_bytes() bytes, row r at
 *           out + r*out_stride.  Exactly the bytes save_mcu_ligne_bw / save_mcu_ligne_color
 *           fwrite after the PPM/PGM header (src/decoder_utils.c:48-90).  out_stride == 0 means
 *           packed rows (the file layout).  Any alignment; at most 2^26 bytes (the kernel reaches a
 *           tile's rows with 32-bit offsets), larger strides return JPEGGPU_ERR_INVALID_ARG.
 *
 * Host-pointer form, synchronous: the frame is cut into stripes of whole MCU rows; all uploads run back to back
 * on one stream, each stripe's kernel and download follow on a second, so that the upload of one stripe, the
 * kernel of the previous and the download of the one before overlap on the full-duplex link (pinned
 * hipMemcpyAsync both ways; see jpeggpu_host_alloc below).  With pageable planes or a
 * pageable raster of 16 MiB and more the call starts a helper thread for that direction's staging copies and
 * joins it before it returns.                                                                              */
int jpeggpu_backend_run(const jpeggpu_frame_desc_t *desc,
                        const jpeggpu_blocl16_t *const coef[3],
                        const jpeggpu_qtable_t *const qtables[4],
                        uint8_t *out, size_t out_stride);

/* Device-pointer form: coef planes and raster are device memory; asynchronous on <hip_stream>
 * (a hipStream_t passed as void*; NULL = the library's own stream).  qtables stay host memory
 * (512 bytes, passed to the kernel by value).                                                  */
int jpeggpu_backend_run_device(const jpeggpu_frame_desc_t *desc,
                               const void *const d_coef[3],
                               const jpeggpu_qtable_t *const qtables[4],
                               void *d_out, size_t out_stride, void *hip_stream);

/* Batch of <nframes> frames of identical geometry in one launch (independent images, the unit
 * that is sharded across GPUs).  Frame f's plane c starts at
 * (char*)d_coef[c] + f*coef_frame_stride[c] bytes, its raster at (char*)d_out + f*out_frame_stride.
 * A stride of 0 means "densely packed".                                                         */
int jpeggpu_backend_run_batch_device(const jpeggpu_frame_desc_t *desc, uint32_t nframes,
                                     const void *const d_coef[3], const size_t coef_frame_stride[3],
                                     const jpeggpu_qtable_t *const qtables[4],
                                     void *d_out, size_t out_stride, size_t out_frame_stride,
                                     void *hip_stream);

/* The batch in host memory (BASELINE.json configs[4] handed over the way the reference holds its planes, src/progressive.c:73-84):
 * synchronous like jpeggpu_backend_run, and ONE pipeline over the stripes of all frames, so the upload of a frame
 * overlaps the kernel and the download of the frame before it (a loop of jpeggpu_backend_run calls cannot: each call
 * ends with its own download).  Strides as above, in bytes, 0 = densely packed.                                       */
int jpeggpu_backend_run_batch(const jpeggpu_frame_desc_t *desc, uint32_t nframes,
                              const jpeggpu_blocl16_t *const coef[3], const size_t coef_frame_stride[3],
                              const jpeggpu_qtable_t *const qtables[4],
                              uint8_t *out, size_t out_stride, size_t out_frame_stride);

/* Blocks until everything queued on <hip_stream> (NULL = library stream) has finished.         */
int jpeggpu_synchronize(void *hip_stream);

/* The same five calls on a context of the caller's (replacing the same reference loop,
 * src/progressive.c:257-312 / src/baseline.c:160-217); hip_stream == NULL means the context's stream.  */
int jpeggpu_ctx_backend_run(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *desc,
                            const jpeggpu_blocl16_t *const coef[3], const jpeggpu_qtable_t *const qtables[4],
                            uint8_t *out, size_t out_stride);
int jpeggpu_ctx_backend_run_batch(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *desc, uint32_t nframes,
                                  const jpeggpu_blocl16_t *const coef[3], const size_t coef_frame_stride[3],
                                  const jpeggpu_qtable_t *const qtables[4],
                                  uint8_t *out, size_t out_stride, size_t out_frame_stride);
int jpeggpu_ctx_backend_run_device(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *desc,
                                   const void *const d_coef[3], const jpeggpu_qtable_t *const qtables[4],
                                   void *d_out, size_t out_stride, void *hip_stream);
int jpeggpu_ctx_backend_run_batch_device(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *desc, uint32_t nframes,
                                         const void *const d_coef[3], const size_t coef_frame_stride[3],
                                         const jpeggpu_qtable_t *const qtables[4],
                                         void *d_out, size_t out_stride, size_t out_frame_stride,
                                         void *hip_stream);
int jpeggpu_ctx_synchronize(jpeggpu_ctx_t *ctx, void *hip_stream);

/* Several GPUs from ONE process (the path shards with no exchange step, so there is no collective).
 * One context per device (several contexts on one device also work); the library runs one host thread per context,
 * each through its context's pipelined host-pointer form, and returns the first error.  No counterpart in the
 * reference (one thread, no device); the unit of work is the reference's own: one MCU row for the writers
 * (src/decoder_utils.c:65-90), one image for the drivers (src/jpeg2ppm.c:62-138).
 *   jpeggpu_multi_backend_run       ONE frame cut into contiguous MCU-row stripes: context g decodes MCU rows
 *                                   [g*nbmcuV/n, (g+1)*nbmcuV/n) of every component and writes the matching band of <out>
 *                                   (replicate upsampling has no halo, src/decoder_utils.c:72-80)
 *   jpeggpu_multi_backend_run_batch <nframes> independent frames of one geometry, host memory, frame f's plane c at
 *                                   (char*)coef[c] + f*coef_frame_stride[c] (0 = dense), raster at out + f*out_frame_stride:
 *                                   frame f goes to context f mod n                                                     */
int jpeggpu_multi_backend_run(jpeggpu_ctx_t *const *ctxs, int nctx, const jpeggpu_frame_desc_t *desc,
                              const jpeggpu_blocl16_t *const coef[3], const jpeggpu_qtable_t *const qtables[4],
                              uint8_t *out, size_t out_stride);
int jpeggpu_multi_backend_run_batch(jpeggpu_ctx_t *const *ctxs, int nctx, const jpeggpu_frame_desc_t *desc, uint32_t nframes,
                                    const jpeggpu_blocl16_t *const coef[3], const size_t coef_frame_stride[3],
                                    const jpeggpu_qtable_t *const qtables[4],
                                    uint8_t *out, size_t out_stride, size_t out_frame_stride);

/* Allocation-free variant of jpeggpu_backend_run (host pointers in and out, synchronous, same bytes).
 * NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot allocate device memory -- code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  Nothing on its path calls hipMalloc,
 * hipFree, hipHostMalloc or any other allocating routine: the device staging that jpeggpu_backend_run grows on
 * demand (coefficient planes in, packed raster out) is a 32 MiB array held statically in the library's device
 * image, one copy on every GPU the library runs on, for the life of the process.
 *   ctx   : a context from jpeggpu_ctx_create, or NULL for the default context, which must already exist
 *           (jpeggpu_init; creating a context allocates, so it is not done here).
 *   limit : sum of the planes (each rounded up to 256 bytes) + rows * row_bytes <= 32 MiB, e.g. one
 *           1920x1080 frame in any sampling; larger frames return JPEGGPU_ERR_INVALID_ARG and touch nothing.
 *   With idct_fast == 0 the context must already hold the "-f" table (one ordinary call in that mode).
 * All calls on one GPU of a process share that GPU's array: the library serialises them, each call overwrites
 * what the last one left there, and nothing else may use the array while a call runs.                        */
int jpeggpu_backend_run_noalloc(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *desc,
                                const jpeggpu_blocl16_t *const coef[3], const jpeggpu_qtable_t *const qtables[4],
                                uint8_t *out, size_t out_stride);

/* Pinned host memory.  jpeggpu_backend_run moves planes and rasters that are pinned -- allocated by
 * jpeggpu_host_alloc / hipHostMalloc (or pinned in place by the caller's own hipHostRegister, where the driver
 * supports user pointers) -- with the DMA engines directly; pageable memory (what the reference's calloc gives,
 * src/progressive.c:73-84) goes through pinned stripe buffers, one extra copy each way.  A binding that
 * lays `sortie` out contiguously allocates it with jpeggpu_host_alloc instead of calloc.
 * The WHOLE range a call touches has to be pinned for the in-place path: the library probes the first and
 * the last byte of every plane and raster it is handed (of every frame of a batch, or an even sample of 256
 * of them) and stages the call through its own buffers otherwise.
 * No counterpart in the reference.                                                                      */
void *jpeggpu_host_alloc(size_t bytes);          /* NULL on failure (jpeggpu_last_error() says why); not zeroed */
void  jpeggpu_host_free(void *ptr);

/* ------------------------------------------------------------------------------------------ */
/* Stage-level array forms of the four function contracts (host pointers, synchronous).        */
/* Same arithmetic and layouts as the per-block reference functions, n items per call.         */
/* ------------------------------------------------------------------------------------------ */

/* iquant, src/iqzz.c:9-14 : blocks[b].data[i] *= q->data[i] for i in [s_start, s_end], int16 wrap. */
int jpeggpu_iquant_blocks(jpeggpu_blocl16_t *blocks, size_t n, uint8_t s_start, uint8_t s_end,
                          const jpeggpu_qtable_t *qtable);
/* izz, src/iqzz.c:16-41 : out[b].data[col][row] = in[b].data[zigzag(row,col)].                 */
int jpeggpu_izz_blocks(const jpeggpu_blocl16_t *in, jpeggpu_bloct16_t *out, size_t n);
/* idct_opt, src/idct_opt.c:93-117.                                                             */
int jpeggpu_idct_opt_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n);
/* idct, src/idct.c:50-71: the slow transform the reference uses under "-f" (direct 64-term sums with
 * calc_coef's table, :35-48).                                                                   */
int jpeggpu_idct_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n);
/* calc_coef's table [x][y][lambda][mu] as this library builds it (host only; cos/sqrt from libm). */
void jpeggpu_idct_coef_table(float *out4096);
/* all_option.idct_fast (src/options.c:76-79, a process-wide option in the reference as well):
 * 1 (default) = idct_opt, 0 = idct for every later jpeggpu_backend_run* / jpeggpu_jpeg_render call on the default
 * context, and the value contexts created later start with.  Atomic; a call reads it once; kept across jpeggpu_shutdown. */
void jpeggpu_set_idct_fast(int fast);
int  jpeggpu_get_idct_fast(void);
/* ycc2rgb_pixel, src/ycc2rgb.c:18-22, over n pixels.                                           */
int jpeggpu_ycc2rgb_pixels(const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                           jpeggpu_rgb_t *rgb, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* JPEGGPU_H */
