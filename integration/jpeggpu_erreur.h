/*
 * jpeggpu_erreur.h -- converts a libjpeggpu status into the reference's erreur_t (include/erreur.h:81-86).
 *
 * Include after the reference's <erreur.h> and <jpeggpu.h>.  The reference returns errors by value:
 * code == SUCCESS (0) on success; `com` is a heap string iff `must_free` (consumer: print_erreur,
 * src/erreur.c:7-10, frees it).  The library keeps its message in a thread-local buffer that the next
 * call overwrites, so the shim hands the caller a heap copy.
 *
 *   0                        -> SUCCESS
 *   > 0                      -> the same erreur_code_t (the library's positive statuses ARE the reference's codes:
 *                               ERR_NB_COMP = 13 from the back end, src/decoder_utils.c:50,67; every header / VLD
 *                               code from jpeggpu_decode.h)
 *   JPEGGPU_ERR_NO_QT        -> ERR_NO_HT: what src/progressive.c:262-267 and src/baseline.c:51-55 return for a
 *                               component without its quantisation table
 *   other negative statuses  -> ERR_PARAM: no device, HIP failure, bad argument, sampling factors the reference's
 *                               integer division cannot handle -- conditions the reference cannot be in; the text says which
 */
#ifndef JPEGGPU_ERREUR_H
#define JPEGGPU_ERREUR_H

#include <stdlib.h>
#include <string.h>

static inline erreur_t jpeggpu_erreur(int rc) {
   if (rc == 0) return (erreur_t) {.code = SUCCESS};
   erreur_code_t code;
   if (rc > 0) code = (erreur_code_t)rc;
   else if (rc == JPEGGPU_ERR_NO_QT) code = ERR_NO_HT;
   else code = ERR_PARAM;
   const char *msg = jpeggpu_last_error();
   size_t n = strlen(msg) + 1;
   char *com = (char *)malloc(n);
   if (com == NULL) return (erreur_t) {.code = code, .com = "libjpeggpu", .must_free = false};
   memcpy(com, msg, n);
   return (erreur_t) {.code = code, .com = com, .must_free = true};
}

#endif /* JPEGGPU_ERREUR_H */
