/*
 * jpeggpu_decode.h -- host front end + output stage around the GPU pixel back end (SURVEY §8f rows
 * f1-f3): JPEG header parser, Huffman VLD into full-frame blocl16_t planes, per-scan hand-off of
 * progressive files, PPM/PGM writer.  The entropy decoder is serial host code (as in the
 * reference); every pixel is still produced by jpeggpu_backend_run* on the GPU.
 *
 * Citations are file:line under the reference tree.  Behaviour is the REFERENCE's, not the
 * standard's, wherever the two differ (listed in DESIGN.md §8): e.g. refinement bits are OR-ed
 * into the coefficient whatever its sign (src/vld.c:177-183), DC scans always walk whole MCUs
 * (src/progressive.c:98-124), only APP0/COM/DQT/DHT/SOF0/SOF2/SOS/EOI markers are accepted
 * (src/entete.c:288-327).
 */
#ifndef JPEGGPU_DECODE_H
#define JPEGGPU_DECODE_H

#include "jpeggpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Positive status codes are erreur_code_t values (include/erreur.h:9-79), so a CLI built on this
 * library exits with the same code as the reference's main (src/jpeg2ppm.c:70-72,93-96,129-133). */
enum {
   JPEGGPU_ERR_0XFF00 = 1, JPEGGPU_ERR_AC_BAD, JPEGGPU_ERR_DC_BAD, JPEGGPU_ERR_SOF_BAD, JPEGGPU_ERR_HUFF_CODE_1,
   JPEGGPU_ERR_DIFF_AH_AL, JPEGGPU_ERR_PARAM, JPEGGPU_ERR_OPT, JPEGGPU_ERR_NO_HT, JPEGGPU_ERR_NO_QT_REF,
   JPEGGPU_ERR_INVALID_FILE_PATH, JPEGGPU_ERR_INVALID_FILE_EXT, /* 13 = JPEGGPU_ERR_NB_COMP */
   JPEGGPU_ERR_SOF_NON_SUPPORTE = 14, JPEGGPU_ERR_APP0_LEN, JPEGGPU_ERR_NO_APP0, JPEGGPU_ERR_COMP_ID, JPEGGPU_ERR_COM_LEN,
   JPEGGPU_ERR_DHT_START_0, JPEGGPU_ERR_DHT_LEN, JPEGGPU_ERR_NO_DHT, JPEGGPU_ERR_DQT_LEN, JPEGGPU_ERR_DQT_PRECISION,
   JPEGGPU_ERR_DQT_ID, JPEGGPU_ERR_NO_DQT, JPEGGPU_ERR_EOI_BEFORE_SOS, JPEGGPU_ERR_NO_EOI, JPEGGPU_ERR_HUFF_BAD,
   JPEGGPU_ERR_HUFF_ID, JPEGGPU_ERR_HUFF_MORE_256, JPEGGPU_ERR_JFIF_VERSION, JPEGGPU_ERR_NO_JFIF, JPEGGPU_ERR_MARKER_BAD,
   JPEGGPU_ERR_MARKER_UNKNOWN, JPEGGPU_ERR_SEVERAL_SOF, JPEGGPU_ERR_SOF_PRECISION, JPEGGPU_ERR_SOF_LEN, JPEGGPU_ERR_NO_SOF,
   JPEGGPU_ERR_SEVERAL_SOI, JPEGGPU_ERR_NO_SOI, JPEGGPU_ERR_SOS_COMP_ID, JPEGGPU_ERR_SOS_SS, JPEGGPU_ERR_SOS_SE,
   JPEGGPU_ERR_SOS_AH, JPEGGPU_ERR_SOS_AL, JPEGGPU_ERR_SOS_NB_COMP, JPEGGPU_ERR_SOS_LEN,
   /* library-specific */
   JPEGGPU_ERR_UNSUPPORTED = -6,   /* input for which the reference itself has no defined behaviour (it reads
                                      uninitialised or NULL data): a baseline scan that lacks a component,
                                      a first scan whose component order is not Y,Cb,Cr with unequal
                                      sampling, coefficient runs that leave the block                     */
   JPEGGPU_ERR_IO = -7
};

typedef struct jpeggpu_jpeg jpeggpu_jpeg_t;

/* What decode_entete left in img_t after the current SOS (include/img.h:88-103, :66-75).        */
typedef struct jpeggpu_scan_info_s {
   uint8_t num_sof;          /* 0 baseline, 2 progressive (section_done_t.num_sof)                */
   uint8_t nb_scan_comp;     /* components named by the SOS                                       */
   uint8_t scan_comp[3];     /* their FRAME indices (get_composante, src/decoder_utils.c:36-46)   */
   uint8_t ss, se, ah, al;   /* other_t                                                           */
   uint8_t eoi;              /* EOI reached: no scan pending                                      */
   uint16_t scans_decoded;   /* scans entropy-decoded so far                                      */
   uint32_t mcu_rows_done;   /* baseline: complete MCU rows of the scan, = nbmcuV unless it failed
                                (the reference has written exactly those rows, src/baseline.c:201-217) */
} jpeggpu_scan_info_t;

/* decode_entete(fichier, true, img) (src/entete.c:199-248): checks SOI/EOI, parses markers up to
 * and including the first SOS, derives the geometry, runs the baseline/progressive header
 * verifications and allocates zeroed full-frame planes (init_sortie, src/progressive.c:73-84, but
 * contiguous).  <data> is borrowed and must outlive the handle.                                  */
int  jpeggpu_jpeg_open_memory(const uint8_t *data, size_t len, jpeggpu_jpeg_t **out);
/* Same, after ouverture_fichier_in's checks (src/utils.c:43-54: extension .jpeg/.jpg, readable). */
int  jpeggpu_jpeg_open_file(const char *path, jpeggpu_jpeg_t **out);
/* Closes the handle.  Its pinned and device buffers are kept for the next file that needs buffers of exactly those
 * sizes (a fresh pinned buffer costs 0.1-0.3 ms per MB when the CPU first touches it), up to JPEGGPU_POOL_BYTES per
 * kind (environment, read once; default 2 GiB, 0 = keep nothing); jpeggpu_shutdown() frees them.                   */
void jpeggpu_jpeg_close(jpeggpu_jpeg_t *j);

int  jpeggpu_jpeg_frame_desc(const jpeggpu_jpeg_t *j, jpeggpu_frame_desc_t *desc);   /* apply_iquant = 1 */
int  jpeggpu_jpeg_scan_info(const jpeggpu_jpeg_t *j, jpeggpu_scan_info_t *info);
/* Quantisation table <idq> as currently defined (NULL if absent); zigzag order like qtable_t.    */
const jpeggpu_qtable_t  *jpeggpu_jpeg_qtable(const jpeggpu_jpeg_t *j, int idq);
/* Host plane of frame component c: QUANTISED zigzag blocks, (blocY*nbH_c + blocX).
 * NOT a pure accessor for progressive files on a GPU host: DC scans keep coefficient 0 of every block in a packed side
 * array (1/64 of the plane: what a DC scan uploads), and this call merges that array into the plane before returning it --
 * it writes through the handle.  So: (1) fetch the pointer AGAIN after every jpeggpu_jpeg_decode_scan (a later DC
 * refinement scan updates only the side array: coefficient 0 behind an old pointer is stale until the next call);
 * (2) like every call on a handle it must not run concurrently with another call on the same handle (two readers
 * would race on the merge).  tests/test_gpu_decode.py::test_plane_pointer_is_refreshed_after_a_dc_refinement_scan.   */
const jpeggpu_blocl16_t *jpeggpu_jpeg_plane(const jpeggpu_jpeg_t *j, int c);
int  jpeggpu_jpeg_nb_comments(const jpeggpu_jpeg_t *j);
const char *jpeggpu_jpeg_comment(const jpeggpu_jpeg_t *j, int i);

/* Entropy-decodes the pending scan into the planes and finishes its last byte:
 * baseline   = the MCU walk of decode_baseline_image (src/baseline.c:160-199) without the pixel
 *              stages (coefficients stay quantised; the GPU applies iquant, same int16 arithmetic),
 * progressive = decode_progressif_dc / decode_progressif_ac + finir_octet (src/progressive.c:98-173,
 *              :220-232).  Host only, no GPU needed.                                              */
int  jpeggpu_jpeg_decode_scan(jpeggpu_jpeg_t *j);
/* decode_entete(fichier, false, img) (src/progressive.c:322): markers up to the next SOS or EOI.  */
int  jpeggpu_jpeg_next_header(jpeggpu_jpeg_t *j);

/* The pixel back end over the current planes (GPU): host raster out, layout of
 * jpeggpu_backend_run.  Planes stay resident on the device between calls; only what the scans since
 * the previous render changed is uploaded again: the plane of a component an AC (or baseline) scan
 * touched, and for a component that only DC scans touched (src/progressive.c:98-126) just the dirty
 * spectral band -- coefficient 0 of every block, 2 of each 128 bytes -- which a small kernel puts in place.
 * A raster in pinned memory (jpeggpu_host_alloc) receives the download in place; any other goes through the
 * handle's own pinned raster and one copy.                                                                  */
int  jpeggpu_jpeg_render(jpeggpu_jpeg_t *j, uint8_t *out, size_t out_stride);
/* Host-to-device bytes this handle has queued so far (a measure of the hand-off, for tests and tuning). */
size_t jpeggpu_jpeg_uploaded_bytes(const jpeggpu_jpeg_t *j);

/* Output stage: "P5\n"|"P6\n", "%d %d\n", "255\n" (src/baseline.c:121-130, src/progressive.c:241-244)
 * followed by ALL raster rows, padded ones included (src/decoder_utils.c:51,68).                 */
int  jpeggpu_write_pnm(const char *path, const jpeggpu_frame_desc_t *desc, const uint8_t *raster, size_t stride);
/* Same with an explicit row count: rows = desc->height gives a file whose payload matches its header
 * (the reference never does that: it always writes the padded rows, src/decoder_utils.c:51,68).      */
int  jpeggpu_write_pnm_rows(const char *path, const jpeggpu_frame_desc_t *desc, const uint8_t *raster, size_t stride, size_t rows);

/* out_file_name (src/utils.c:56-90): <outfile or input path> with "-<nb>" inserted before the
 * extension when nb != 0; without outfile the extension becomes .pgm/.ppm.  Returns the length
 * written (excluding NUL) or a negative status.                                                   */
int  jpeggpu_out_file_name(const char *filepath, const char *outfile, uint8_t nbcomp, uint8_t nb, char *buf, size_t cap);

/* The whole of main() for one file (src/jpeg2ppm.c:62-141): baseline -> one image; progressive ->
 * one image per scan, "<name>-<k>.<ext>", and a symbolic link "<name>.<ext>" to the last one
 * (src/progressive.c:332-343).  *nb_images receives the number of images written.                */
int  jpeggpu_jpeg2ppm(const char *filepath, const char *outfile, int *nb_images);

/* The stages the reference times under -t (print_timer, src/timer.c:36-45; call sites src/jpeg2ppm.c:84-94,
 * :134-135, src/baseline.c:224-228, src/progressive.c:316-321), for this library's decomposition of the work.
 * Host wall-clock seconds.  gpu_wait_s is the time the host thread spent enqueuing uploads/launches and
 * waiting for rasters, write_s the time it waited for the thread that writes the files: GPU work and file
 * writing overlap the entropy decoding (of the next scans of a progressive file, of the following MCU rows of
 * a baseline file), so the parts do not add up to the GPU's or the writer's busy time.                  */
typedef struct jpeggpu_timing_s {
   double init_s;      /* binding the GPU (HIP runtime start-up, stream): first call of the process only        */
   double header_s;    /* "Décodage entête" + the headers between scans                                   */
   double vld_s;       /* "Décodage DC/AC et Quantification" (entropy decoding; iquant runs on the GPU; a baseline
                          file's hand-over of finished rows, some tens of microseconds each, is inside)         */
   double gpu_wait_s;  /* upload + kernel + download as seen by the host thread (IZZ + IDCT + colour)      */
   double write_s;     /* "Ecriture de l'image": what of it the host thread had to wait for                 */
   double total_s;     /* "Temps total"                                                                   */
} jpeggpu_timing_t;
/* jpeggpu_jpeg2ppm with the stage times; timing may be NULL.                                             */
int  jpeggpu_jpeg2ppm_timed(const char *filepath, const char *outfile, int *nb_images, jpeggpu_timing_t *timing);

/* ------------------------------------------------------------------------------------------ */
/* Batches of independent images (BASELINE.json configs[4]: thousands of same-size files)        */
/* ------------------------------------------------------------------------------------------ */

typedef struct jpeggpu_blob_s { const uint8_t *data; size_t len; } jpeggpu_blob_t;

typedef struct jpeggpu_batch_stats_s {
   double seconds;            /* wall time of the call                                            */
   double vld_seconds;        /* entropy decoding, summed over the host threads                   */
   uint32_t images_ok;        /* images with status 0                                             */
   uint32_t threads;          /* host threads used                                                */
   uint32_t chunk_frames;     /* frames per GPU launch                                            */
   uint32_t chunks;           /* launches                                                         */
   uint32_t output_registered;/* 1: rasters pinned in place (no staging copy)                     */
} jpeggpu_batch_stats_t;

/* Decodes <n> JPEG files of identical geometry (width, height, components, sampling factors; tables
 * may differ from file to file) into rasters + i*raster_frame_stride (each laid out like the output of
 * jpeggpu_backend_run; stride 0 = packed).  No counterpart in the reference, which decodes one file per
 * process; every image goes through the same front end and the same kernel as jpeggpu_jpeg2ppm (for
 * progressive files only the image after the last scan is produced).
 *
 * Host threads run the header parser and Huffman VLD of different images in parallel, writing
 * dequantised coefficients (iquant as in decode_bloc_baseline, src/baseline.c:66) straight into a pinned
 * chunk buffer; each chunk is one hipMemcpyAsync, one jpeggpu_backend_run_batch_device launch
 * (apply_iquant = 0) and one copy back, double-buffered so that the GPU works on chunk k while the
 * threads decode chunk k+1.  status[i] (may be NULL) receives each image's status; a failed image
 * leaves its raster undefined.  Returns 0 when the batch ran (even if some images failed), a negative
 * status when it could not.  nthreads <= 0: min(16, hardware threads).                             */
int jpeggpu_decode_batch(const jpeggpu_blob_t *jpegs, size_t n, uint8_t *rasters, size_t raster_frame_stride,
                         int nthreads, int *status, jpeggpu_frame_desc_t *desc_out, jpeggpu_batch_stats_t *stats);

#ifdef __cplusplus
}
#endif
#endif /* JPEGGPU_DECODE_H */
