// jpeg2ppm_gpu -- thin command-line driver over jpeggpu_jpeg2ppm: same arguments for input and
// output file as the reference's jpeg2ppm (src/options.c:58-66: "-o <fichier>" / "--outfile=<fichier>")
// and the same exit status (the erreur_code_t of the failure, src/jpeg2ppm.c:70-72,93-96,129-133).
// "-f" / "--no-fast-idct" selects the slow transform like the reference; "-t" / "--timer" prints the stage
// times in print_timer's format (src/timer.c:43) for this library's stages.  Everything else of the
// reference's CLI (verbose, table dumps) is out of scope.
#include <cstdio>
#include <cstring>

#include "../../include/jpeggpu.h"
#include "../../include/jpeggpu_decode.h"

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define JPEG2PPM_ASAN 1
extern "C" void __sanitizer_purge_allocator(void);
#endif
#endif

static int usage(const char *exe, int code, const char *why) {
   if (why) fprintf(stderr, "Erreur %d : %s\n", code, why);
   fprintf(stderr, "Usage : %s [-t | --timer] [-f | --no-fast-idct] [-o <fichier> | --outfile=<fichier>] fichier.jpg|fichier.jpeg\n", exe);
   return code;
}

static int run(int argc, char **argv);

// Everything the library holds on the GPU (default context, streams, events, the "-f" table, the pools of pinned and
// device buffers kept between files) is given back BEFORE main returns, not left to static destruction after it: the HIP
// and HSA runtimes tear themselves down from their own finalisers, in an order this program does not control.
int main(int argc, char **argv) {
   const int rc = run(argc, argv);
   jpeggpu_shutdown();
#if defined(JPEG2PPM_ASAN)
   // Host-ASan builds (tools/asan_cli.sh): ROCm's AddressSanitizer runtime keeps freed HSA allocations (pinned and device
   // buffers: it wraps hsa_amd_memory_pool_allocate) in its quarantine and gives them back to HSA when the quarantine is
   // recycled -- if that happens inside the runtimes' finalisers, after HSA has shut down, it aborts with "CHECK failed:
   // sanitizer_allocator_device.h ... dev_runtime_unloaded_".  Drain the quarantine now, while the runtime is alive.
   __sanitizer_purge_allocator();
#endif
   return rc;
}

static int run(int argc, char **argv) {
   const char *in = nullptr, *out = nullptr;
   bool timer = false;
   for (int i = 1; i < argc; i++) {
      const char *a = argv[i];
      if (a[0] != '-') {
         if (in) return usage(argv[0], JPEGGPU_ERR_PARAM, "Deux images passées en paramètre.");
         in = a;
      } else if (strcmp(a, "-h") == 0 || strcmp(a, "--help") == 0) {
         usage(argv[0], 0, nullptr);
         return 0;
      } else if (strcmp(a, "-t") == 0 || strcmp(a, "--timer") == 0) {
         timer = true;               // set_timer_param, src/options.c:71-74
      } else if (strcmp(a, "-f") == 0 || strcmp(a, "--no-fast-idct") == 0) {
         jpeggpu_set_idct_fast(0);   // set_idct_nofast_param, src/options.c:76-79
      } else if (strcmp(a, "-o") == 0) {
         if (i + 1 >= argc) return usage(argv[0], JPEGGPU_ERR_OPT, "Manque la valeur pour le paramètre 'o'");
         if (out) return usage(argv[0], JPEGGPU_ERR_PARAM, "Maximum une image en output.");
         out = argv[++i];
      } else if (strncmp(a, "--outfile=", 10) == 0) {
         if (a[10] == 0) return usage(argv[0], JPEGGPU_ERR_PARAM, "Manque la valeur pour le paramètre 'outfile'");
         if (out) return usage(argv[0], JPEGGPU_ERR_PARAM, "Maximum une image en output.");
         out = a + 10;
      } else {
         return usage(argv[0], JPEGGPU_ERR_PARAM, "Pas de paramètre de ce nom");
      }
   }
   if (!in) return usage(argv[0], JPEGGPU_ERR_INVALID_FILE_PATH, "Pas de fichier jpeg/jpg");
   int n = 0;
   jpeggpu_timing_t tm;
   int rc = jpeggpu_jpeg2ppm_timed(in, out, &n, &tm);
   if (timer) {
      fprintf(stdout, "%s : %f s\n", "Initialisation du GPU", (float)tm.init_s);
      fprintf(stdout, "%s : %f s\n", "Décodage entête", (float)tm.header_s);
      fprintf(stdout, "%s : %f s\n", "Décodage DC/AC", (float)tm.vld_s);
      fprintf(stdout, "%s : %f s\n", "GPU (quantification, IZZ, IDCT, couleur) vu de l'hôte", (float)tm.gpu_wait_s);
      fprintf(stdout, "%s : %f s\n", "Ecriture de l'image", (float)tm.write_s);
      fprintf(stdout, "%s : %f s\n", "Temps total", (float)tm.total_s);
   }
   if (rc) {
      fprintf(stderr, "Erreur %d : %s\n", rc, jpeggpu_last_error());   // print_erreur, src/erreur.c:7-10
      return rc > 0 ? rc : 1;
   }
   return 0;
}
