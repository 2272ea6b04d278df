// frontend.cpp -- see frontend.hpp.  Every function names the reference lines whose behaviour it
// reproduces; the code itself is written for a memory-resident file and table-driven decoding.
#include "frontend.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(__x86_64__) && (defined(__GNUC__) || defined(__clang__))
#include <immintrin.h>
#define JFE_HAVE_BMI2_PATH 1
#endif

#include "../../include/jpeggpu_decode.h"

namespace jpegfe {

static_assert(JPEGGPU_ERR_NO_HT == 9 && JPEGGPU_ERR_SOF_NON_SUPPORTE == 14 && JPEGGPU_ERR_NO_EOI == 27 &&
              JPEGGPU_ERR_MARKER_UNKNOWN == 34 && JPEGGPU_ERR_SOS_LEN == 47, "erreur_code_t numbering (include/erreur.h:9-79)");

int Decoder::fail(int code, const char *fmt, ...) {
   char b[256];
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(b, sizeof b, fmt, ap);
   va_end(ap);
   err = b;
   return code;
}

// fgetc semantics: past the end every read yields EOF, which the reference stores in a uint8_t (0xFF)
int Decoder::getc() { return pos < len ? data[pos++] : 0xFF; }
int Decoder::get16() { int a = getc(); return ((a << 8) + getc()) & 0xFFFF; }

// ---------------------------------------------------------------------------------------------
// Huffman tables
// ---------------------------------------------------------------------------------------------

// remplir_huffman (src/entete.c:452-507).  Nodes are visited breadth first; a node at the depth of the
// next code length takes the next symbol, any other node gets two children.  The table is refused
// (ERR_HUFF_BAD) when no node is left over at the end, i.e. when the all-ones code would be in use.
// An over-subscribed length list makes the reference expand nodes without end; it is refused too.
int HuffTable::build(const uint8_t counts[16], const uint8_t *symbols, unsigned nsym) {
   nodes.clear();
   nodes.push_back(Node{ { -1, -1 }, 0 });
   uint8_t lens[256];
   unsigned n = 0;
   for (unsigned l = 1; l <= 16; l++)
      for (unsigned k = 0; k < counts[l - 1] && n < 256; k++) lens[n++] = (uint8_t)l;
   if (nsym >= 256) return JPEGGPU_ERR_HUFF_BAD;   // the reference's 8-bit symbol counter wraps and never terminates
   // level by level: `level` holds the nodes of the current depth, in left-to-right order
   std::vector<int> level{ 0 }, next;
   unsigned num = 0;
   bool leftover = false;
   for (unsigned depth = 0; num < nsym; depth++) {
      if (level.empty()) return JPEGGPU_ERR_HUFF_BAD;
      if (lens[num] < depth) return JPEGGPU_ERR_HUFF_BAD;   // more codes of a length than nodes: over-subscribed
      next.clear();
      size_t i = 0;
      for (; i < level.size() && num < nsym; i++) {
         int nd = level[i];
         if (lens[num] == depth) {
            nodes[nd].sym = symbols[num++];
         } else {
            int c0 = (int)nodes.size();
            nodes.push_back(Node{ { -1, -1 }, 0 });
            nodes.push_back(Node{ { -1, -1 }, 0 });
            nodes[nd].child[0] = c0;
            nodes[nd].child[1] = c0 + 1;
            next.push_back(c0);
            next.push_back(c0 + 1);
         }
      }
      if (num == nsym) { leftover = i < level.size() || !next.empty(); break; }
      level.swap(next);
   }
   if (nsym == 0) leftover = true;    // the queue still holds the root
   if (!leftover) return JPEGGPU_ERR_HUFF_BAD;
   if (nodes.size() > ((size_t)1 << 17)) return JPEGGPU_ERR_HUFF_BAD;   // cannot happen: depth <= 16 bounds the tree at 2^17 - 1 nodes
   // lookup tables
   for (unsigned w = 0; w < 512; w++) {
      int nd = 0, l = 0;
      while (l < 9 && nodes[nd].child[0] >= 0) { nd = nodes[nd].child[(w >> (8 - l)) & 1]; l++; }
      if (nodes[nd].child[0] < 0 && l > 0) { fast[w] = (uint16_t)((l << 8) | nodes[nd].sym); deep[w] = 0; }
      else { fast[w] = 0; deep[w] = nd; }
   }
   for (unsigned w = 0; w < 1024; w++) {
      acfast[w] = 0;
      const unsigned f = fast[w >> 1];
      if (!f) continue;
      const unsigned len = f >> 8, sym = f & 0xFF, g = sym & 15;
      if (g == 0 || g > 10 || len + g > 10) continue;
      const unsigned v = (w >> (10 - len - g)) & ((1u << g) - 1);
      const int lo = 1 << (g - 1);
      const int val = (int)v < lo ? (int)v - ((lo << 1) - 1) : (int)v;   // get_val_from_magnitude, src/vld.c:75-86
      acfast[w] = (int32_t)(((uint32_t)val << 16) | ((sym >> 4) << 8) | (len + g));
   }
   present = true;
   return 0;
}

// ---------------------------------------------------------------------------------------------
// Bit-level primitives
// ---------------------------------------------------------------------------------------------

// Two ways to read the scan's bits, with one interface, so that every decoder below exists once (as a template):
//   CheckedBits  the bit position and its end, as the reference's read_bit sees them: any read can run into the
//                marker (ok() turns false).  Used where the scan is about to end.
//   FastBits     the next bits in a register, topped up from memory by a branch-free refill; ok() is always true.
//                Used for a block only when the scan still holds more bits than any block can consume
//                (kBlockBitsMargin), so that the reference's end-of-data errors cannot occur inside it.
// window(): the next bits, first bit in bit 63, at least 56 of them valid.
struct CheckedBits {
   static constexpr bool kChecked = true;
   BitReader &br;
   explicit CheckedBits(BitReader &b) : br(b) {}
   uint64_t window() const {
      uint64_t v;
      memcpy(&v, br.buf + (br.pos >> 3), 8);
      return __builtin_bswap64(v) << (br.pos & 7);
   }
   void skip(unsigned n) { br.pos += n; }
   bool ok() const { return br.pos < br.limit; }
   uint64_t pos() const { return br.pos; }
   uint64_t limit() const { return br.limit; }
   bool bit(uint32_t &b) {                                    // read_bit, src/bitstream.c:13-23
      b = (uint32_t)(window() >> 63);
      br.pos += 1;
      return br.pos < br.limit;
   }
   void done() {}
};

struct FastBits {
   static constexpr bool kChecked = false;
   BitReader &br;
   const uint8_t *p;
   uint64_t bits = 0;
   unsigned n = 0;                                            // valid bits at the top of <bits>; what lies below them are the stream's next bits as well
   explicit FastBits(BitReader &b) : br(b), p(b.buf + (b.pos >> 3)) {
      refill();
      bits <<= (b.pos & 7);
      n -= (unsigned)(b.pos & 7);
   }
   void refill() {
      uint64_t w;
      memcpy(&w, p, 8);
      bits |= __builtin_bswap64(w) >> n;
      const unsigned adv = (63 - n) >> 3;
      p += adv;
      n += adv * 8;                                           // 56..63
   }
   uint64_t window() { refill(); return bits; }
   void skip(unsigned k) { bits <<= k; n -= k; }
   static constexpr bool ok() { return true; }
   uint64_t pos() const { return 8 * (uint64_t)(p - br.buf) - n; }
   uint64_t limit() const { return br.limit; }
   bool bit(uint32_t &b) {
      if (n == 0) refill();
      b = (uint32_t)(bits >> 63);
      bits <<= 1;
      n -= 1;
      return true;
   }
   void done() { br.pos = pos(); }                            // hand the position back to the scan's BitReader
};
constexpr uint64_t kBlockBitsMargin = 2048;   // > 1 + 63 symbols of 16 + 11 bits (1728); > 63 x (16 + 1) + 63 correction bits + an end-of-band run
std::atomic<int> g_vld_mode{ 0 };   // test hook (jpeggpu_internal_vld_force_checked): 1 = every block through CheckedBits, 2 = no BMI2 code
#if defined(JFE_HAVE_BMI2_PATH)
static bool cpu_has_bmi2() { static const bool v = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt"); return v; }
#endif

static inline bool far_from_end(const BitReader &br) { return br.fast_allowed && br.pos + kBlockBitsMargin < br.limit; }

// get_huffman_symbole (src/vld.c:268-283).  Returns the symbol, or -1 when the walk crossed the end
// of the scan data (read_bit failing on the byte after 0xFF, src/bitstream.c:13-23), or -2 for a code no
// symbol was given to.  *all_ones: the bits read were all 1 (code_que_un).
template <class R>
static inline int huff_symbol(R &r, const HuffTable &t, bool *all_ones) {
   const uint64_t start = r.pos();
   const uint32_t w = (uint32_t)(r.window() >> 48);
   unsigned f = t.fast[w >> 7], len;
   int sym;
   if (f) {
      len = f >> 8;
      sym = (int)(f & 0xFF);
   } else {
      int nd = t.deep[w >> 7];
      len = 9;
      if (t.nodes[nd].child[0] < 0) return -2;   // empty table: the reference dereferences NULL
      while (t.nodes[nd].child[0] >= 0 && len < 16) { nd = t.nodes[nd].child[(w >> (15 - len)) & 1]; len++; }
      if (t.nodes[nd].child[0] >= 0) return -2;
      sym = t.nodes[nd].sym;
   }
   r.skip(len);
   if (!r.ok()) {
      if (all_ones) {   // bits actually read before the failure: start .. limit-1
         unsigned nb = (unsigned)(r.limit() - start);
         *all_ones = (w >> (16 - nb)) == ((1u << nb) - 1);
      }
      return -1;
   }
   if (all_ones) *all_ones = (w >> (16 - len)) == ((1u << len) - 1);
   return sym;
}

// read_indice (src/vld.c:88-100): n bits, MSB first.  false when the scan data ran out.
template <class R>
static inline bool read_index(R &r, unsigned n, uint32_t &idx) {
   idx = 0;
   if (n == 0) return true;
   idx = (uint32_t)(r.window() >> (64 - n));
   r.skip(n);
   return r.ok();
}

// get_val_from_magnitude (src/vld.c:75-86)
static inline int magnitude_value(unsigned m, uint32_t idx) {
   if (m == 0) return 0;
   int lo = 1 << (m - 1);
   return (int)idx < lo ? (int)idx - ((lo << 1) - 1) : (int)idx;
}

static const char *const kMsgFF = "Pas de 0x00 après un 0xff (Pas bien !!)";

// ---------------------------------------------------------------------------------------------
// Header segments
// ---------------------------------------------------------------------------------------------

int Decoder::seg_app0() {   // app0, src/entete.c:330-346
   if (get16() != 16) return fail(JPEGGPU_ERR_APP0_LEN, "[APP0] Longueur section APP0 incorrecte");
   for (int i = 0; i < 5; i++) jfif[i] = (char)getc();
   ver_x = (uint8_t)getc();
   ver_y = (uint8_t)getc();
   pos += 7;
   app0_done = true;
   return 0;
}

int Decoder::seg_com() {   // com, src/entete.c:349-365
   int length = get16();
   if (length < 2) return fail(JPEGGPU_ERR_COM_LEN, "[COM] Longueur section COM incorrecte");
   std::string s;
   for (int i = 0; i < length - 2 && pos < len; i++) s.push_back((char)data[pos++]);
   comments.push_back(s);
   return 0;
}

int Decoder::seg_sof() {   // sof, src/entete.c:368-404
   if (sof_done) return fail(JPEGGPU_ERR_SEVERAL_SOF, "[SOF] Plusieurs SOF");
   int length = get16();
   precision = (uint8_t)getc();
   height = (uint16_t)get16();
   width = (uint16_t)get16();
   int nb = getc();
   if (length != 8 + 3 * nb) return fail(JPEGGPU_ERR_SOF_LEN, "[SOF] Longueur section SOF incorrecte");
   nbcomp = (uint8_t)nb;
   for (int i = 0; i < nb; i++) {
      int id = getc();
      if (id == 0) return fail(JPEGGPU_ERR_COMP_ID, "[SOF] Indice composante doit être différent de 0");
      int samp = getc(), idq = getc();
      if (i < 3) {
         comp[i].idc = (uint8_t)id;
         comp[i].hs = (uint8_t)(samp >> 4);
         comp[i].vs = (uint8_t)(samp & 15);
         comp[i].idq = (uint8_t)idq;
      } else {
         too_many_comps = true;   // the reference writes past comps[3] here (include/img.h:50)
      }
   }
   sof_done = true;
   return 0;
}

int Decoder::seg_dqt() {   // dqt, src/entete.c:407-449
   int length = get16();
   if ((length - 2) % 65 != 0) return fail(JPEGGPU_ERR_DQT_LEN, "[DQT] Longueur section DQT incorrecte");
   for (int n = 1; n <= (length - 2) / 65; n++) {
      int octet = getc();
      int prec = octet >> 4;
      if (prec != 0 && prec != 1)
         return fail(JPEGGPU_ERR_DQT_PRECISION, "[DQT] Précision table de quantification doit valoir 0 ou 1 (8 bits ou 16 bits)");
      int id = octet & 15;
      if (id > 3) return fail(JPEGGPU_ERR_DQT_ID, "[DQT] Indice table de quantification doit être entre 0 et 3");
      qt_present[id] = true;
      qt_precision[id] = (uint8_t)prec;
      for (int i = 0; i < 64; i++) qt[id][i] = (uint16_t)(prec == 0 ? getc() : get16());
   }
   dqt_done = true;
   return 0;
}

int Decoder::seg_dht() {   // dht, src/entete.c:510-578
   const size_t begin = pos;
   const size_t length = (size_t)get16();
   while (pos < begin + length) {
      if (pos >= len) return fail(JPEGGPU_ERR_DHT_START_0, "[DHT] 3 premiers bits de la section DHT doivent valoir 0");
      int octet = getc();
      if (octet & 0xE0) return fail(JPEGGPU_ERR_DHT_START_0, "[DHT] 3 premiers bits de la section DHT doivent valoir 0");
      bool is_dc = (octet & 0x10) == 0;
      int id = octet & 15;
      if (id > 3) return fail(JPEGGPU_ERR_HUFF_ID, "[DHT] Indice table de Huffman doit être entre 0 et 3");
      uint8_t counts[16];
      unsigned nb = 0;
      for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)getc(); nb += counts[i]; }
      if (nb > 256) return fail(JPEGGPU_ERR_HUFF_MORE_256, "[DHT] Plus de 256 symboles dans la table de Huffman");
      uint8_t symbols[256];
      for (unsigned i = 0; i < nb; i++) symbols[i] = (uint8_t)getc();
      HuffTable &t = is_dc ? hdc[id] : hac[id];
      t.present = false;
      if (t.build(counts, symbols, nb)) { t.present = true; return fail(JPEGGPU_ERR_HUFF_BAD, "[DHT] Table Huffman incorrecte"); }
   }
   if (pos != begin + length) return fail(JPEGGPU_ERR_DHT_LEN, "[DHT] Longueur section DHT incorrecte");
   dht_done = true;
   return 0;
}

int Decoder::seg_sos() {   // verif_sos + sos, src/entete.c:581-645
   if (!app0_done) return fail(JPEGGPU_ERR_NO_APP0, "Image sans APP0 (ou SOS avant APP0)");
   if (!sof_done) return fail(JPEGGPU_ERR_NO_SOF, "Image sans SOF (ou SOS avant SOF)");
   if (!dqt_done) return fail(JPEGGPU_ERR_NO_DQT, "Image sans DQT (ou SOS avant DQT)");
   if (!dht_done) return fail(JPEGGPU_ERR_NO_DHT, "Image sans DHT (ou SOS avant DHT)");
   int length = get16();
   int nb = getc();
   if (nb > nbcomp) return fail(JPEGGPU_ERR_SOS_NB_COMP, "[SOS] Nombre de composantes dans le SOS supérieur au nombre total");
   if (length != 6 + 2 * nb) return fail(JPEGGPU_ERR_SOS_LEN, "[SOS] Longueur section SOS incorrecte");
   ordre[0] = ordre[1] = ordre[2] = 0;
   nb_scan_comp = (uint8_t)nb;
   const int nframe = nbcomp > 3 ? 3 : nbcomp;
   for (int i = 0; i < nb; i++) {
      int id = getc(), tb = getc();
      if (i < 3) ordre[i] = (uint8_t)id;
      int j = 0;
      while (j < nframe && comp[j].idc != id) j++;
      if (j >= nframe) return fail(JPEGGPU_ERR_SOS_COMP_ID, "[SOS] Indice de composante incorrect");
      comp[j].idhdc = (uint8_t)(tb >> 4);
      comp[j].idhac = (uint8_t)(tb & 15);
   }
   ss = (uint8_t)getc();
   se = (uint8_t)getc();
   int a = getc();
   ah = (uint8_t)(a >> 4);
   al = (uint8_t)(a & 15);
   sos_done = true;
   return 0;
}

int Decoder::marker() {   // marqueur, src/entete.c:275-327
   if (pos + 2 > len) return fail(JPEGGPU_ERR_MARKER_BAD, "Octet 0x%zx devrait être un marqueur : fin de fichier", pos);
   int m0 = data[pos], m1 = data[pos + 1];
   pos += 2;
   if (m0 != 0xFF) return fail(JPEGGPU_ERR_MARKER_BAD, "Octet 0x%zx devrait être un marqueur : %x %x", pos - 2, m0, m1);
   switch (m1) {
   case 0xC0: num_sof = 0; return seg_sof();
   case 0xC2: num_sof = 2; return seg_sof();
   case 0xC4: return seg_dht();
   case 0xD8: return fail(JPEGGPU_ERR_SEVERAL_SOI, "Plusieurs SOI");
   case 0xD9: eoi_done = true; return 0;
   case 0xDA: return seg_sos();
   case 0xDB: return seg_dqt();
   case 0xE0: return seg_app0();
   case 0xFE: return seg_com();
   default: return fail(JPEGGPU_ERR_MARKER_UNKNOWN, "Marqueur inconnu : %x", m1);
   }
}

int Decoder::check_app0() {   // verif_entete_app0, src/entete.c:60-72
   if (memcmp(jfif, "JFIF", 5) != 0) return fail(JPEGGPU_ERR_NO_JFIF, "[APP0] Phrase JFIF manquante dans APP0");
   if (ver_x != 1) return fail(JPEGGPU_ERR_JFIF_VERSION, "[APP0] Version JFIF X doit valoir 1");
   if (ver_y != 1) return fail(JPEGGPU_ERR_JFIF_VERSION, "[APP0] Version JFIF Y doit valoir 1");
   return 0;
}

int Decoder::check_baseline() {   // verif_entete_baseline, src/entete.c:75-133
   if (precision != 8) return fail(JPEGGPU_ERR_SOF_PRECISION, "[SOF0] Précision composante doit valoir 8 (Baseline)");
   for (int i = 0; i < 4; i++)
      if (qt_present[i] && qt_precision[i] != 0)
         return fail(JPEGGPU_ERR_DQT_PRECISION, "[DQT] Précision table de quantification doit valoir 0 (8 bits) (Baseline)");
   if (hdc[2].present || hdc[3].present) return fail(JPEGGPU_ERR_HUFF_ID, "[DHT] Indice table de Huffman DC doit valoir 0 ou 1");
   if (hac[2].present || hac[3].present) return fail(JPEGGPU_ERR_HUFF_ID, "[DHT] Indice table de Huffman AC doit valoir 0 ou 1");
   for (int i = 0; i < 3 && i < nbcomp; i++) {
      if (comp[i].idhdc > 1) return fail(JPEGGPU_ERR_HUFF_ID, "[SOS] Indice table de Huffman DC doit valoir 0 ou 1 (Baseline)");
      if (comp[i].idhac > 1) return fail(JPEGGPU_ERR_HUFF_ID, "[SOS] Indice table de Huffman AC doit valoir 0 ou 1 (Baseline)");
   }
   if (ss != 0) return fail(JPEGGPU_ERR_SOS_SS, "[SOS] Ss doit valoir 0 (Baseline)");
   if (se != 63) return fail(JPEGGPU_ERR_SOS_SE, "[SOS] Se doit valoir 63 (Baseline)");
   if (ah != 0) return fail(JPEGGPU_ERR_SOS_AH, "[SOS] Ah doit valoir 0 (Baseline)");
   if (al != 0) return fail(JPEGGPU_ERR_SOS_AL, "[SOS] Al doit valoir 0 (Baseline)");
   return 0;
}

int Decoder::check_progressive() {   // verif_entete_progressif, src/entete.c:136-173
   if (precision != 8)
      return fail(JPEGGPU_ERR_SOF_PRECISION, precision == 12 ? "[SOF2] Précision composante 12 non pris charge (Progressif)"
                                                             : "[SOF2] Précision composante invalide (Progressif)");
   for (int i = 0; i < 3 && i < nbcomp; i++)
      if (comp[i].idc > 4) return fail(JPEGGPU_ERR_COMP_ID, "[SOF2] Indice composante doit valoir entre 1 et 4 (Progressif)");
   if (ss > 63) return fail(JPEGGPU_ERR_SOS_SS, "[SOS] Ss doit valoir entre 0 et 63 (Progressif)");
   if (se < ss || se > 63) return fail(JPEGGPU_ERR_SOS_SE, "[SOS] Se doit valoir entre Ss et 63 (Progressif)");
   if (ah > 13) return fail(JPEGGPU_ERR_SOS_AH, "[SOS] Ah doit valoir entre 0 et 13 (Progressif)");
   if (al > 13) return fail(JPEGGPU_ERR_SOS_AL, "[SOS] Al doit valoir entre 0 et 13 (Progressif)");
   return 0;
}

void Decoder::geometry() {   // calcul_image_information, src/entete.c:176-197 (float ceil of float quotients)
   int nbh = (int)std::ceil((float)width / 8), nbv = (int)std::ceil((float)height / 8);
   max_h = max_v = 0;
   for (int i = 0; i < nbcomp && i < 3; i++) {
      if (comp[i].hs > max_h) max_h = comp[i].hs;
      if (comp[i].vs > max_v) max_v = comp[i].vs;
   }
   nbmcuH = max_h ? (uint32_t)std::ceil((float)nbh / max_h) : 0;
   nbmcuV = max_v ? (uint32_t)std::ceil((float)nbv / max_v) : 0;
}

int Decoder::header_loop(bool first) {   // decode_entete, src/entete.c:199-248
   if (first) {
      if (len < 2 || data[0] != 0xFF || data[1] != 0xD8) return fail(JPEGGPU_ERR_NO_SOI, "L'image doit commencer par 0xffd8 (SOI)");
      if (data[len - 2] != 0xFF || data[len - 1] != 0xD9) return fail(JPEGGPU_ERR_NO_EOI, "L'image doit finir par 0xffd9 (EOI)");
      pos = 2;
   } else {
      sos_done = false;
   }
   while (!sos_done && !eoi_done) {
      int rc = marker();
      if (rc) return rc;
   }
   if (sos_done) {
      if (first) {
         geometry();
         int rc = check_app0();
         if (rc) return rc;
         if (num_sof == 0 && (rc = check_baseline())) return rc;
      }
      if (num_sof == 2) {
         int rc = check_progressive();
         if (rc) return rc;
      }
   } else if (eoi_done && first) {
      return fail(JPEGGPU_ERR_EOI_BEFORE_SOS, "Image sans image");
   }
   return 0;
}

int Decoder::open() {
   int rc = header_loop(true);
   if (rc) return rc;
   // What the reference's drivers need before touching the stream and where it has no defined
   // behaviour of its own.
   if (too_many_comps || (nbcomp != 1 && nbcomp != 3))
      return fail(JPEGGPU_ERR_NB_COMP, "Il faut une ou trois composante");   // src/baseline.c:127-128
   for (int c = 0; c < nbcomp; c++) {
      if (comp[c].hs == 0 || comp[c].vs == 0 || comp[c].hs > 4 || comp[c].vs > 4 || max_h % comp[c].hs || max_v % comp[c].vs)
         return fail(JPEGGPU_ERR_SAMPLING, "composante %d : facteurs %ux%u incompatibles avec %ux%u", c, comp[c].hs, comp[c].vs, max_h, max_v);
   }
   if (nbcomp == 3) {
      // get_ycc_info (src/decoder_utils.c:92-110) derives the plane of Y/Cb/Cr from the FIRST scan's
      // component order; it is only coherent when that scan names all three components and, unless
      // every component has the same sampling, in frame order.
      bool all = nb_scan_comp == 3, ident = true, same = true;
      for (int i = 0; i < 3; i++) {
         if (ordre[i] != comp[i].idc) ident = false;
         if (comp[i].hs != comp[0].hs || comp[i].vs != comp[0].vs) same = false;
      }
      if (!all || (!ident && !(same && num_sof == 0)))
         return fail(JPEGGPU_ERR_UNSUPPORTED, "premier scan sans les trois composantes dans l'ordre Y Cb Cr : comportement indéfini dans la référence");
   } else if (num_sof == 0 && nb_scan_comp != 1) {
      return fail(JPEGGPU_ERR_UNSUPPORTED, "scan baseline sans composante");
   }
   return 0;
}

int Decoder::next_header() { return header_loop(false); }

int Decoder::get_composante(unsigned k) const {   // src/decoder_utils.c:36-46
   if (k >= 3) return -1;
   uint8_t id = ordre[k];
   if (id == 0) return -1;
   for (int c = 0; c < nbcomp && c < 3; c++)
      if (comp[c].idc == id) return c;
   return -1;
}

// ---------------------------------------------------------------------------------------------
// Scan data
// ---------------------------------------------------------------------------------------------

// init_bitstream + my_getc (src/bitstream.c:13-31): the entropy-coded segment runs up to the first 0xFF
// that is not followed by 0x00; every "FF 00" stands for a data byte FF.  The segment is un-stuffed
// once; the marker's own FF is appended because the reference hands out its bits too and only fails
// when the eighth of them has been read.
int Decoder::begin_scan(BitReader &br, size_t &scan_begin, size_t &scan_end, size_t &nbytes) {
   scan_begin = pos;
   scanbuf.clear();
   size_t p = pos;
   while (true) {
      const uint8_t *ff = p < len ? (const uint8_t *)memchr(data + p, 0xFF, len - p) : nullptr;
      size_t q = ff ? (size_t)(ff - data) : len;
      scanbuf.insert(scanbuf.end(), data + p, data + q);
      if (!ff) { p = len; break; }                     // unreachable: the file ends with FF D9
      if (q + 1 < len && data[q + 1] == 0x00) { scanbuf.push_back(0xFF); p = q + 2; continue; }
      p = q;
      break;
   }
   scan_end = p;
   nbytes = scanbuf.size();
   scanbuf.push_back(0xFF);
   scanbuf.insert(scanbuf.end(), 24, 0);
   br.buf = scanbuf.data();
   br.pos = 0;
   br.limit = 8 * (uint64_t)nbytes + 8;
   br.fast_allowed = g_vld_mode.load() != 1;
#if defined(JFE_HAVE_BMI2_PATH)
   br.bmi2 = cpu_has_bmi2() && g_vld_mode.load() == 0;
#endif
   return 0;
}

// finir_octet (src/bitstream.c:45-60): leaves the file position on the byte after the one the last bit
// came from (un-reading the look-ahead byte when the scan stopped on a byte boundary).
int Decoder::finish_scan(const BitReader &br, size_t scan_begin, size_t scan_end, size_t nbytes) {
   const uint64_t idx = br.pos >> 3;
   const unsigned off = (unsigned)(br.pos & 7);
   if (off != 0 && idx >= nbytes) return fail(JPEGGPU_ERR_0XFF00, "%s %zx", kMsgFF, scan_end + 1);
   uint64_t want = off == 0 ? idx : idx + 1;            // un-stuffed bytes in front of the new position
   if (want >= nbytes) { pos = scan_end; return 0; }    // the usual case: the next thing is the marker
   size_t p = scan_begin;                               // rare (trailing bytes in the segment): walk the stuffing
   for (uint64_t k = 0; k < want; k++) p += (data[p] == 0xFF) ? 2 : 1;
   pos = p;
   return 0;
}

// decode_coef_DC, first scan of the band (src/vld.c:285-305 + :102-113)
template <class R>
static inline int dc_first(R &br, const HuffTable &t, unsigned al, int16_t *coef, std::string &err) {
   bool ones = true;
   int s = huff_symbol(br, t, &ones);
   if (s == -2) { err = "[DHT] table de Huffman DC vide"; return JPEGGPU_ERR_HUFF_BAD; }
   if (ones) { err = "Le code de huffman avec que des 1 est utilisé"; return JPEGGPU_ERR_HUFF_CODE_1; }
   if (s < 0) { err = kMsgFF; return JPEGGPU_ERR_0XFF00; }
   if (s > 11) { err = "La magnitude doit être inférieur ou égale à 11"; return JPEGGPU_ERR_DC_BAD; }
   uint32_t idx;
   if (!read_index(br, (unsigned)s, idx)) { err = kMsgFF; return JPEGGPU_ERR_0XFF00; }
   *coef = (int16_t)(magnitude_value((unsigned)s, idx) * (1 << al));
   return 0;
}

// decode_coef_AC_first_scan in a loop (src/vld.c:124-174, :307-327).  One window per coefficient
// serves the Huffman code (<= 16 bits) and the value bits (<= 10) that follow it.
template <class R>
int Decoder::ac_first(R &r, const HuffTable &t, int16_t *blk, uint16_t &skip, const uint16_t *q) {
   unsigned idx = ss;
   skip = 0;
   const unsigned last = se;
   while (idx <= last) {
      const uint64_t w = r.window();                         // >= 56 valid bits
      if (const int32_t e = t.acfast[w >> 54]) {             // code and value bits within 10 bits: run, length and value in one look-up
         r.skip((unsigned)e & 0xFF);
         if (!r.ok()) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
         idx += ((unsigned)e >> 8) & 15;
         if (idx > 63) return fail(JPEGGPU_ERR_UNSUPPORTED, "coefficient AC hors du bloc (indice %u)", idx);
         int c = (e >> 16) * (1 << al);
         if (q) c = (int16_t)c * (int)q[idx];
         blk[idx] = (int16_t)c;
         idx++;
         continue;
      }
      const unsigned f = t.fast[w >> 55];
      unsigned clen;
      int s;
      if (f) {
         clen = f >> 8;
         s = (int)(f & 0xFF);
      } else {
         const uint32_t w16 = (uint32_t)(w >> 48);
         int nd = t.deep[w >> 55];
         clen = 9;
         if (t.nodes[nd].child[0] < 0) return fail(JPEGGPU_ERR_HUFF_BAD, "[DHT] table de Huffman AC vide");
         while (t.nodes[nd].child[0] >= 0 && clen < 16) { nd = t.nodes[nd].child[(w16 >> (15 - clen)) & 1]; clen++; }
         if (t.nodes[nd].child[0] >= 0) return fail(JPEGGPU_ERR_HUFF_BAD, "[DHT] table de Huffman AC vide");
         s = t.nodes[nd].sym;
      }
      r.skip(clen);
      if (!r.ok()) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
      if (s == 0xF0) { idx += 16; continue; }
      const unsigned alpha = (unsigned)s >> 4, gamma = (unsigned)s & 15;
      if (gamma == 0) {
         if (alpha == 0) { skip = 1; return 0; }
         if (num_sof == 0) return fail(JPEGGPU_ERR_AC_BAD, "Code invalide pour AC (%x) car mode baseline", s);
         if (alpha > 14) return fail(JPEGGPU_ERR_AC_BAD, "EOB%d interdit (max = 14)", alpha);
         uint32_t run;
         if (!read_index(r, alpha, run)) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
         skip = (uint16_t)(run + (1u << alpha));
         return 0;
      }
      idx += alpha;
      if (gamma > 10) return fail(JPEGGPU_ERR_AC_BAD, "La magnitude doit être inférieur ou égale à 10");
      const uint32_t v = (uint32_t)((w << clen) >> (64 - gamma));   // clen + gamma <= 26 bits of the window
      r.skip(gamma);
      if (!r.ok()) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
      if (idx > 63) return fail(JPEGGPU_ERR_UNSUPPORTED, "coefficient AC hors du bloc (indice %u)", idx);   // the reference writes past the block
      int c = magnitude_value(gamma, v) * (1 << al);
      if (q) c = (int16_t)c * (int)q[idx];                         // iquant, src/iqzz.c:12 (only non-zero coefficients change)
      blk[idx] = (int16_t)c;
      idx++;
   }
   return 0;
}

// Refinement scans walk a block's coefficients one by one and branch on "is it zero" -- a coin toss for the branch
// predictor.  Away from the end of the scan (FastBits) the walk is replaced by arithmetic on a 64-bit map of the
// block's non-zero coefficients: the zero that ends a run is found by clearing set bits, and the coefficients that
// receive a correction bit on the way are the map's set bits in between, visited in ascending order as the walk does.
static inline uint64_t nonzero_mask(const int16_t *blk) {   // bit i: blk[i] != 0
#if defined(__SSE2__)
   const __m128i zero = _mm_setzero_si128();
   uint64_t m = 0;
   for (int k = 0; k < 4; k++) {
      const __m128i a = _mm_cmpeq_epi16(_mm_loadu_si128((const __m128i *)(blk + 16 * k)), zero);
      const __m128i b = _mm_cmpeq_epi16(_mm_loadu_si128((const __m128i *)(blk + 16 * k + 8)), zero);
      m |= (uint64_t)(uint16_t)_mm_movemask_epi8(_mm_packs_epi16(a, b)) << (16 * k);   // bit i: blk[i] == 0
   }
   return ~m;
#else
   uint64_t m = 0;
   for (int i = 0; i < 64; i++) m |= (uint64_t)(blk[i] != 0) << i;
   return m;
#endif
}

// correction_coef (src/vld.c:177-183) for every position in <m>, ascending: the next bit, OR-ed in at bit <al> whatever the sign
static inline void correct_positions(FastBits &r, int16_t *blk, uint64_t m, unsigned al) {
   while (m) {
      const unsigned i = (unsigned)__builtin_ctzll(m);
      m &= m - 1;
      uint32_t b;
      r.bit(b);
      blk[i] = (int16_t)(blk[i] | (int16_t)(b << al));
   }
}

// The correction bits of one block, gathered as a bit map (bit i: coefficient i receives 1 << al) and OR-ed in at the end
static inline void apply_corrections(int16_t *blk, uint64_t corr, unsigned al) {
#if defined(__SSE2__)
   const __m128i sel = _mm_setr_epi16(1, 2, 4, 8, 16, 32, 64, 128);
   const __m128i bitv = _mm_set1_epi16((short)(1 << al));
   for (int g = 0; g < 8; g++) {
      const unsigned byte = (unsigned)(corr >> (8 * g)) & 0xFF;
      if (!byte) continue;
      __m128i v = _mm_and_si128(_mm_set1_epi16((short)byte), sel);
      v = _mm_and_si128(_mm_cmpeq_epi16(v, sel), bitv);
      __m128i *p = (__m128i *)(blk + 8 * g);
      _mm_storeu_si128(p, _mm_or_si128(_mm_loadu_si128(p), v));
   }
#else
   for (; corr; corr &= corr - 1) { const unsigned i = (unsigned)__builtin_ctzll(corr); blk[i] = (int16_t)(blk[i] | (int16_t)(1 << al)); }
#endif
}

#if defined(JFE_HAVE_BMI2_PATH)
static inline uint64_t bitrev64(uint64_t x) {
   x = __builtin_bswap64(x);
   x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
   x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
   x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
   return x;
}

// One stream bit for every set bit of <m>, the first bit for the lowest position (the order of the reference's walk);
// returned at those positions.  PDEP hands the low bits of its source to the set bits of the mask in ascending order,
// so the window is mirrored first (its first bit is bit 63).
__attribute__((target("bmi2,popcnt")))
static inline uint64_t take_corrections(FastBits &r, uint64_t m) {
   const unsigned c = (unsigned)__builtin_popcountll(m);
   if (__builtin_expect(c > 56, 0)) {                         // more bits than one window holds: the two halves of the block separately
      const uint64_t lo = take_corrections(r, m & 0xFFFFFFFFull);
      return lo | take_corrections(r, m & ~0xFFFFFFFFull);
   }
   const uint64_t got = _pdep_u64(bitrev64(r.window()), m);
   r.skip(c);
   return got;
}

// decode_coef_AC_subsequent_scan for one block without a branch per coefficient: the zero that ends a run is selected
// with PDEP, the correction bits on the way are deposited into a bit map and OR-ed into the block once
__attribute__((target("bmi2,popcnt")))
int Decoder::ac_refine_bmi2(FastBits &br, const HuffTable &t, int16_t *blk, uint16_t &skip) {
   unsigned idx = ss;
   const uint64_t nz = nonzero_mask(blk);
   uint64_t corr = 0;
   int rc = 0;
   while (idx <= se) {
      const uint8_t *const p0 = br.p;
      const uint64_t bits0 = br.bits;
      const unsigned n0 = br.n;
      const int s = huff_symbol(br, t, nullptr);
      if (s < 0) { rc = fail(JPEGGPU_ERR_HUFF_BAD, "[DHT] table de Huffman AC vide"); break; }
      const uint64_t from = ~0ull << idx;
      const unsigned alpha = (unsigned)s >> 4, gamma = (unsigned)s & 15;
      if (gamma == 0 && s != 0xF0) {
         if (alpha == 0) {
            skip = 1;
         } else {
            uint32_t run;
            read_index(br, alpha, run);
            skip = (uint16_t)(run + (1u << alpha));
         }
         corr |= take_corrections(br, nz & from & (~0ull >> (63 - se)));
         break;
      }
      if (s != 0xF0 && gamma != 1) {
         rc = fail(JPEGGPU_ERR_AC_BAD, "En progressif les AC qui ne sont pas sur le premier scan doivent être 0xRRRRSSSS avec SSSS=0 ou 1");
         break;
      }
      uint32_t sb = 0;
      if (s != 0xF0) br.bit(sb);
      const uint64_t sel = _pdep_u64(1ull << (s == 0xF0 ? 15 : alpha), ~nz & from);   // the 16th / (run + 1)-th zero from here
      if (!sel) {
         br.p = p0; br.bits = bits0; br.n = n0;
         apply_corrections(blk, corr, al);
         return ac_refine_walk(br, t, blk, skip, idx);
      }
      const unsigned T = (unsigned)__builtin_ctzll(sel);
      corr |= take_corrections(br, nz & from & ((1ull << T) - 1));
      if (s != 0xF0) blk[T] = (int16_t)((sb ? 1 : -1) * (1 << al));
      idx = T + 1;
   }
   apply_corrections(blk, corr, al);
   return rc;
}

__attribute__((target("bmi2,popcnt")))
static void refine_band_bmi2(FastBits &br, int16_t *blk, uint64_t band, unsigned al) {
   apply_corrections(blk, take_corrections(br, nonzero_mask(blk) & band), al);
}
#endif

// correction_eob (src/vld.c:228-238): one correction bit for every non-zero coefficient up to Se
template <class R>
int Decoder::refine_to_band_end(R &br, int16_t *blk, unsigned idx) {
   if constexpr (!R::kChecked) {                              // no read can fail: visit the non-zero coefficients only
      if (idx > se) return 0;
      const uint64_t band = (~0ull << idx) & (~0ull >> (63 - se));
#if defined(JFE_HAVE_BMI2_PATH)
      if (br.br.bmi2) { refine_band_bmi2(br, blk, band, al); return 0; }
#endif
      correct_positions(br, blk, nonzero_mask(blk) & band, al);
      return 0;
   }
   for (; idx <= se; idx++) {
      if (blk[idx] != 0) {
         uint32_t b;
         if (!br.bit(b)) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
         if (b) blk[idx] = (int16_t)(blk[idx] | (int16_t)(1 << al));   // correction_coef, src/vld.c:177-183: OR whatever the sign
      }
   }
   return 0;
}

// decode_coef_AC_subsequent_scan in a loop (src/vld.c:185-266, :307-327)
template <class R>
int Decoder::ac_refine(R &br, const HuffTable &t, int16_t *blk, uint16_t &skip) {
   skip = 0;
   if constexpr (R::kChecked) {
      return ac_refine_walk(br, t, blk, skip, ss);
   } else {
#if defined(JFE_HAVE_BMI2_PATH)
      if (br.br.bmi2) return ac_refine_bmi2(br, t, blk, skip);
#endif
      unsigned idx = ss;
      const uint64_t nz = nonzero_mask(blk);                  // coefficients this block places itself lie below idx and are never met again
      while (idx <= se) {
         const uint8_t *const p0 = br.p;                      // the reader in front of this symbol: anything out of the ordinary
         const uint64_t bits0 = br.bits;                      // is handed to the coefficient-by-coefficient walk from here
         const unsigned n0 = br.n;
         const int s = huff_symbol(br, t, nullptr);
         if (s < 0) return fail(JPEGGPU_ERR_HUFF_BAD, "[DHT] table de Huffman AC vide");
         const uint64_t from = ~0ull << idx;                  // idx <= se <= 63
         const unsigned alpha = (unsigned)s >> 4, gamma = (unsigned)s & 15;
         if (gamma == 0 && s != 0xF0) {                       // end of band (run)
            if (alpha == 0) {
               skip = 1;
            } else {
               uint32_t run;
               read_index(br, alpha, run);
               skip = (uint16_t)(run + (1u << alpha));
            }
            correct_positions(br, blk, nz & from & (~0ull >> (63 - se)), al);
            return 0;
         }
         if (s != 0xF0 && gamma != 1)
            return fail(JPEGGPU_ERR_AC_BAD, "En progressif les AC qui ne sont pas sur le premier scan doivent être 0xRRRRSSSS avec SSSS=0 ou 1");
         uint32_t sb = 0;
         if (s != 0xF0) br.bit(sb);                           // the sign comes before the correction bits
         // ZRL: the walk stops behind the 16th zero.  (run, 1): behind <run> zeros it goes on to the next zero, which
         // takes the new coefficient -- the (run + 1)-th zero from here.
         uint64_t z = ~nz & from;
         for (unsigned k = (s == 0xF0 ? 15 : alpha); k != 0 && z; k--) z &= z - 1;
         if (!z) {                                            // the block has not that many zeros left: the reference runs out of it
            br.p = p0; br.bits = bits0; br.n = n0;
            return ac_refine_walk(br, t, blk, skip, idx);
         }
         const unsigned T = (unsigned)__builtin_ctzll(z);
         correct_positions(br, blk, nz & from & ((1ull << T) - 1), al);
         if (s != 0xF0) blk[T] = (int16_t)((sb ? 1 : -1) * (1 << al));
         idx = T + 1;
      }
      return 0;
   }
}

// ... and the same, coefficient by coefficient as the reference does it (src/vld.c:185-266), from coefficient <idx> on
template <class R>
int Decoder::ac_refine_walk(R &br, const HuffTable &t, int16_t *blk, uint16_t &skip, unsigned idx) {
   const int16_t bitv = (int16_t)(1 << al);
   // correction_n_coef (src/vld.c:185-198): pass <n> zero coefficients, correcting the non-zero ones met
   auto pass_zeros = [&](unsigned n) -> int {
      unsigned z = 0;
      while (z < n) {
         if (idx > 63) return fail(JPEGGPU_ERR_UNSUPPORTED, "plage de zéros hors du bloc");   // the reference reads past the block
         if (blk[idx] != 0) {
            uint32_t b;
            if (!br.bit(b)) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
            if (b) blk[idx] = (int16_t)(blk[idx] | bitv);
         } else {
            z++;
         }
         idx++;
      }
      return 0;
   };
   while (idx <= se) {
      int s = huff_symbol(br, t, nullptr);
      if (s < 0) return s == -2 ? fail(JPEGGPU_ERR_HUFF_BAD, "[DHT] table de Huffman AC vide") : fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
      if (s == 0xF0) {
         int rc = pass_zeros(16);
         if (rc) return rc;
         continue;
      }
      unsigned alpha = (unsigned)s >> 4, gamma = (unsigned)s & 15;
      if (gamma == 0) {
         if (alpha == 0) {
            skip = 1;
         } else {
            uint32_t run;
            if (!read_index(br, alpha, run)) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
            skip = (uint16_t)(run + (1u << alpha));
         }
         return refine_to_band_end(br, blk, idx);
      }
      if (gamma != 1)
         return fail(JPEGGPU_ERR_AC_BAD, "En progressif les AC qui ne sont pas sur le premier scan doivent être 0xRRRRSSSS avec SSSS=0 ou 1");
      // skip_n_coef_AC_subsequent_scan (src/vld.c:212-222): sign bit first, then the zero run, then the
      // non-zero coefficients up to the next zero, which receives +-(1<<al)
      uint32_t sb;
      if (!br.bit(sb)) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
      int rc = pass_zeros(alpha);
      if (rc) return rc;
      while (true) {
         if (idx > 63) return fail(JPEGGPU_ERR_UNSUPPORTED, "nouveau coefficient hors du bloc");
         if (blk[idx] == 0) break;
         uint32_t b;
         if (!br.bit(b)) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
         if (b) blk[idx] = (int16_t)(blk[idx] | bitv);
         idx++;
      }
      blk[idx] = (int16_t)((sb ? 1 : -1) * (1 << al));
      idx++;
   }
   return 0;
}

// One whole block of a baseline scan (decode_bloc_acdc_baseline, src/vld.c:329-343, with Al = 0) for the case that
// nothing is wrong with it.  The caller has made sure that the scan holds more bits than any block can take (1 DC +
// 63 AC symbols of at most 16 + 11 bits), so no read can run into the marker; the bits sit in a register that is
// topped up from memory once per symbol, which keeps the address arithmetic of the checked path (bit position ->
// byte load -> swap -> shift) off the dependency chain from one symbol to the next.  Anything the reference treats
// as an error, or that this path does not handle (an unused code, a magnitude out of range, a run past the block, an
// end-of-band run), returns false: the caller decodes the block again from the same bit position by the checked
// path, which stores the same coefficients up to that point and then reports what the reference reports.
template <bool Q>   // Q: store coefficient * q[index] (iquant at store time, src/iqzz.c:12)
__attribute__((always_inline)) static inline bool baseline_block_fast(const uint8_t *buf, uint64_t &pos_io, const HuffTable &tdc, const HuffTable &tac,
                                       int16_t *blk, int16_t pred, int16_t &dc_out, const uint16_t *q) {
   const uint8_t *p = buf + (pos_io >> 3);
   uint64_t bits = 0;
   unsigned n = 0;                        // valid bits at the top of <bits>; the bits below them are the stream's next ones too
#define JFE_REFILL()                                                           \
   do {                                                                        \
      uint64_t w_;                                                             \
      memcpy(&w_, p, 8);                                                       \
      bits |= __builtin_bswap64(w_) >> n;                                      \
      const unsigned adv_ = (63 - n) >> 3;                                     \
      p += adv_;                                                               \
      n += adv_ * 8;                       /* 56..63 */                        \
   } while (0)
   JFE_REFILL();
   bits <<= (pos_io & 7);
   n -= (unsigned)(pos_io & 7);
   JFE_REFILL();
   // DC: decode_coef_DC (src/vld.c:285-305)
   {
      unsigned f = tdc.fast[bits >> 55], len, s;
      if (f) {
         len = f >> 8;
         s = f & 0xFF;
      } else {
         const uint32_t w16 = (uint32_t)(bits >> 48);
         int nd = tdc.deep[bits >> 55];
         len = 9;
         if (tdc.nodes[nd].child[0] < 0) return false;
         while (tdc.nodes[nd].child[0] >= 0 && len < 16) { nd = tdc.nodes[nd].child[(w16 >> (15 - len)) & 1]; len++; }
         if (tdc.nodes[nd].child[0] >= 0) return false;
         s = tdc.nodes[nd].sym;
      }
      if ((bits >> (64 - len)) == ((1ull << len) - 1)) return false;   // the all-ones code
      if (s > 11) return false;
      bits <<= len;
      n -= len;
      int v = 0;
      if (s) {
         const uint32_t idx = (uint32_t)(bits >> (64 - s));
         bits <<= s;
         n -= s;
         v = magnitude_value(s, idx);
      }
      const int16_t dc = (int16_t)((int16_t)v + pred);
      dc_out = dc;
      blk[0] = Q ? (int16_t)(dc * (int)q[0]) : dc;
   }
   // AC: decode_coef_AC_first_scan, Ss = 1, Se = 63 (src/vld.c:124-174)
   unsigned idx = 1;
   while (idx <= 63) {
      JFE_REFILL();
      const int32_t e = tac.acfast[bits >> 54];
      if (e) {                                                  // code + value within 10 bits
         const unsigned tl = (unsigned)e & 0xFF;
         idx += ((unsigned)e >> 8) & 15;
         if (idx > 63) return false;
         bits <<= tl;
         n -= tl;
         const int c = e >> 16;
         blk[idx] = Q ? (int16_t)((int16_t)c * (int)q[idx]) : (int16_t)c;
         idx++;
         continue;
      }
      unsigned f = tac.fast[bits >> 55], len, s;
      if (f) {
         len = f >> 8;
         s = f & 0xFF;
      } else {
         const uint32_t w16 = (uint32_t)(bits >> 48);
         int nd = tac.deep[bits >> 55];
         len = 9;
         if (tac.nodes[nd].child[0] < 0) return false;
         while (tac.nodes[nd].child[0] >= 0 && len < 16) { nd = tac.nodes[nd].child[(w16 >> (15 - len)) & 1]; len++; }
         if (tac.nodes[nd].child[0] >= 0) return false;
         s = tac.nodes[nd].sym;
      }
      bits <<= len;
      n -= len;
      if (s == 0xF0) { idx += 16; continue; }
      const unsigned alpha = s >> 4, gamma = s & 15;
      if (gamma == 0) {
         if (alpha == 0) break;                                 // end of block
         return false;
      }
      if (gamma > 10) return false;
      idx += alpha;
      if (idx > 63) return false;
      const uint32_t v = (uint32_t)(bits >> (64 - gamma));
      bits <<= gamma;
      n -= gamma;
      const int c = magnitude_value(gamma, v);
      blk[idx] = Q ? (int16_t)((int16_t)c * (int)q[idx]) : (int16_t)c;
      idx++;
   }
#undef JFE_REFILL
   pos_io = 8 * (uint64_t)(p - buf) - n;
   return true;
}

// The MCU walk of decode_baseline_image (src/baseline.c:160-199) with decode_bloc_acdc_baseline
// (src/vld.c:329-343); blocks go to their frame position instead of through izz/idct.
int Decoder::scan_baseline(BitReader &br) {
   int16_t pred[3] = { 0, 0, 0 };
   int scan_c[3], nsc = 0;
   for (unsigned k = 0; k < nbcomp; k++) {
      int c = get_composante(k);
      if (c < 0) break;
      scan_c[nsc++] = c;
   }
   if (nsc != nbcomp) return fail(JPEGGPU_ERR_UNSUPPORTED, "scan baseline sans toutes les composantes");
   const uint8_t ss0 = ss;
   // the unchecked block decoder serves the regular case: Al = 0, the band 1..63 (the reference decodes Ss = 1 .. Se
   // whatever the header says only as far as Se, src/vld.c:329-343) and enough bits left for any block
   const bool fast_ok = al == 0 && se == 63;
   mcu_rows_done = 0;
   for (uint32_t my = 0; my < nbmcuV; my++) {
      for (uint32_t mx = 0; mx < nbmcuH; mx++) {
         for (int k = 0; k < nsc; k++) {
            const int c = scan_c[k];
            const Component &ci = comp[c];
            const size_t nbH = (size_t)nbmcuH * ci.hs;
            for (unsigned by = 0; by < ci.vs; by++)
               for (unsigned bx = 0; bx < ci.hs; bx++) {
                  // decode_bloc_baseline's table checks (src/baseline.c:42-57)
                  if (ci.idhdc > 3 || !hdc[ci.idhdc].present) return fail(JPEGGPU_ERR_NO_HT, "Pas de table de huffman DC pour la composante %d", c);
                  if (ci.idhac > 3 || !hac[ci.idhac].present) return fail(JPEGGPU_ERR_NO_HT, "Pas de table de huffman AC pour la composante %d", c);
                  if (ci.idq > 3 || !qt_present[ci.idq]) return fail(JPEGGPU_ERR_NO_HT, "Pas de table de quantification pour la composante %d", c);
                  int16_t *blk = plane[c] + (((size_t)my * ci.vs + by) * nbH + (size_t)mx * ci.hs + bx) * 64;
                  const uint16_t *q = dequant_on_store ? qt[ci.idq] : nullptr;
                  if (fast_ok && far_from_end(br)) {
                     int16_t dc;
                     const bool done = q ? baseline_block_fast<true>(br.buf, br.pos, hdc[ci.idhdc], hac[ci.idhac], blk, pred[c], dc, q)
                                         : baseline_block_fast<false>(br.buf, br.pos, hdc[ci.idhdc], hac[ci.idhac], blk, pred[c], dc, q);
                     if (done) { pred[c] = dc; continue; }
                  }
                  CheckedBits cb(br);
                  int rc = dc_first(cb, hdc[ci.idhdc], al, blk, err);
                  if (rc) return rc;
                  blk[0] = (int16_t)(blk[0] + pred[c]);
                  pred[c] = blk[0];
                  if (q) blk[0] = (int16_t)(blk[0] * (int)q[0]);
                  uint16_t skip;
                  ss = 1;
                  rc = ac_first(cb, hac[ci.idhac], blk, skip, q);
                  ss = ss0;
                  if (rc) return rc;
               }
         }
      }
      mcu_rows_done = my + 1;
      if (row_hook && row_hook_every && mcu_rows_done % row_hook_every == 0 && mcu_rows_done != nbmcuV) {
         const int hr = row_hook(row_hook_ctx, mcu_rows_done);
         if (hr) return hr;
      }
   }
   for (int k = 0; k < nsc; k++) { dirty[scan_c[k]] = true; dirty_kind[scan_c[k]] = kWhole; }
   return 0;
}

// decode_progressif_dc (src/progressive.c:98-126) with the DC branch of decode_bloc_acdc_progressif
// (src/vld.c:345-361): whole MCUs whatever the number of components in the scan.
int Decoder::scan_progressive_dc(BitReader &br) {
   int16_t pred[3] = { 0, 0, 0 };
   int scan_c[3], nsc = 0;
   for (unsigned k = 0; k < nbcomp; k++) {
      int c = get_composante(k);
      if (c < 0) break;
      scan_c[nsc++] = c;
   }
   if (ss != 0) return fail(JPEGGPU_ERR_UNSUPPORTED, "scan AC avec Se = 0");   // unreachable: Se >= Ss is verified
   for (uint32_t my = 0; my < nbmcuV; my++)
      for (uint32_t mx = 0; mx < nbmcuH; mx++)
         for (int k = 0; k < nsc; k++) {
            const int c = scan_c[k];
            const Component &ci = comp[c];
            const size_t nbH = (size_t)nbmcuH * ci.hs;
            for (unsigned by = 0; by < ci.vs; by++)
               for (unsigned bx = 0; bx < ci.hs; bx++) {
                  if (ci.idhdc > 3 || !hdc[ci.idhdc].present)   // src/progressive.c:53-57
                     return fail(JPEGGPU_ERR_NO_HT, "Pas de table de huffman DC pour la composante %d", c);
                  const size_t bi = ((size_t)my * ci.vs + by) * nbH + (size_t)mx * ci.hs + bx;
                  const bool packed_only = dc_in_pack && dc_pack[c];
                  int16_t *const dc = packed_only ? dc_pack[c] + bi : plane[c] + bi * 64;   // where this block's coefficient 0 lives
                  if (ah != 0) {
                     if (ah - al != 1) return fail(JPEGGPU_ERR_DIFF_AH_AL, "La différence entre ah et al devrait être 1");
                     CheckedBits cb(br);
                     uint32_t b;
                     if (!cb.bit(b)) return fail(JPEGGPU_ERR_0XFF00, "%s", kMsgFF);
                     *dc = (int16_t)(*dc | (int16_t)(b << al));   // src/vld.c:115-122
                  } else {
                     int rc;
                     if (far_from_end(br)) {
                        FastBits fb(br);
                        rc = dc_first(fb, hdc[ci.idhdc], al, dc, err);
                        fb.done();
                     } else {
                        CheckedBits cb(br);
                        rc = dc_first(cb, hdc[ci.idhdc], al, dc, err);
                     }
                     if (rc) return rc;
                     *dc = (int16_t)(*dc + pred[c]);
                  }
                  pred[c] = *dc;
                  if (dc_pack[c] && !packed_only) dc_pack[c][bi] = *dc;
               }
         }
   for (int k = 0; k < nsc; k++) {
      dirty[scan_c[k]] = true;
      dc_dirty[scan_c[k]] = true;
      if (dirty_kind[scan_c[k]] < kDcOnly) dirty_kind[scan_c[k]] = kDcOnly;
   }
   return 0;
}

// Coefficient 0 of every block from the packed arrays into the planes (dc_in_pack): for callers that read the planes.
void Decoder::merge_dc_into_planes(int only) {
   if (!dc_in_pack) return;
   for (int c = 0; c < nbcomp && c < 3; c++) {
      if ((only >= 0 && c != only) || !dc_pack[c] || !plane[c]) continue;
      const size_t nb = plane_blocks(c);
      for (size_t b = 0; b < nb; b++) plane[c][b * 64] = dc_pack[c][b];
   }
}

// decode_progressif_ac (src/progressive.c:128-173): the first component of the scan only, blocks in
// raster order over the un-padded block grid, end-of-band runs carried from block to block.
int Decoder::scan_progressive_ac(BitReader &br) {
   const int c = get_composante(0);
   if (c < 0) return fail(JPEGGPU_ERR_COMP_ID, "Aucune composante dans le scan");
   if (ss == 0)   // the reference dereferences a NULL predictor here (src/vld.c:352 with dc_prec == NULL)
      return fail(JPEGGPU_ERR_UNSUPPORTED, "scan progressif avec Ss = 0 et Se != 0");
   const Component &ci = comp[c];
   const uint64_t nb_blocH = (uint64_t)std::ceil((double)width / 8), nb_blocV = (uint64_t)std::ceil((double)height / 8);
   const unsigned hf = max_h / ci.hs, vf = max_v / ci.vs;
   const size_t nb_totalH = (size_t)nbmcuH * ci.hs;
   const uint64_t nbH = (uint64_t)std::ceil((double)nb_blocH / hf), nbV = (uint64_t)std::ceil((double)nb_blocV / vf);
   uint16_t skip = 0;
   for (uint64_t by = 0; by < nbV; by++)
      for (uint64_t bx = 0; bx < nbH; bx++) {
         int16_t *blk = plane[c] + ((size_t)by * nb_totalH + (size_t)bx) * 64;
         if (skip == 0) {
            if (ci.idhac > 3 || !hac[ci.idhac].present)   // src/progressive.c:58-62 (Se != 0 here)
               return fail(JPEGGPU_ERR_NO_HT, "Pas de table de huffman AC pour la composante %d", c);
            if (ah != 0 && ah - al != 1) return fail(JPEGGPU_ERR_DIFF_AH_AL, "La différence entre ah et al devrait être 1");
            int rc;
            if (far_from_end(br)) {       // far from the end of the scan: no read of this block can fail
               FastBits fb(br);
               rc = ah == 0 ? ac_first(fb, hac[ci.idhac], blk, skip, nullptr) : ac_refine(fb, hac[ci.idhac], blk, skip);
               fb.done();
            } else {
               CheckedBits cb(br);
               rc = ah == 0 ? ac_first(cb, hac[ci.idhac], blk, skip, nullptr) : ac_refine(cb, hac[ci.idhac], blk, skip);
            }
            if (rc) return rc;
            if (skip != 0) skip--;   // src/progressive.c:68
         } else {
            if (ah != 0) {
               int rc;
               if (far_from_end(br)) {
                  FastBits fb(br);
                  rc = refine_to_band_end(fb, blk, ss);
                  fb.done();
               } else {
                  CheckedBits cb(br);
                  rc = refine_to_band_end(cb, blk, ss);
               }
               if (rc) return rc;
            }
            skip--;
         }
      }
   dirty[c] = true;
   dirty_kind[c] = kWhole;
   return 0;
}

void Decoder::dequantise_planes() {
   for (int c = 0; c < nbcomp && c < 3; c++) {
      if (!plane[c] || comp[c].idq > 3 || !qt_present[comp[c].idq]) continue;
      const uint16_t *q = qt[comp[c].idq];
      int16_t *p = plane[c];
      const size_t nb = plane_blocks(c);
      for (size_t b = 0; b < nb; b++, p += 64)
         for (int i = 0; i < 64; i++) p[i] = (int16_t)(p[i] * (int)q[i]);
   }
}

int Decoder::decode_scan() {
   if (eoi_done || !sos_done) return fail(JPEGGPU_ERR_INVALID_ARG, "pas de scan en attente");
   for (int c = 0; c < nbcomp; c++)
      if (!plane[c]) return fail(JPEGGPU_ERR_INVALID_ARG, "plans de coefficients non attachés");
   BitReader br;
   size_t b, e, n;
   begin_scan(br, b, e, n);
   int rc;
   if (num_sof == 0) rc = scan_baseline(br);
   else if (se == 0) rc = scan_progressive_dc(br);   // src/progressive.c:220-221
   else rc = scan_progressive_ac(br);
   if (rc) return rc;
   rc = finish_scan(br, b, e, n);
   if (rc) return rc;
   scans_decoded++;
   sos_done = false;
   if (num_sof == 0) eoi_done = true;   // decode_baseline_image returns here: nothing after the scan is ever read (src/baseline.c:219-231)
   return 0;
}

}  // namespace jpegfe

// Test hook.  mode 1: decode every block with the reader that checks for the end of the scan on every read (the form that
// mirrors the reference's read_bit line by line); tests compare it with the default (0), in which only the last 256 bytes
// of a scan are decoded that way.  mode 2: the default without the BMI2 form of the refinement scans.
extern "C" void jpeggpu_internal_vld_force_checked(int mode) { jpegfe::g_vld_mode.store(mode); }
