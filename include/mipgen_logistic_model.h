/*
 * mipgen_logistic_model.h — parameter table of MIPgen's logistic scorer, as data.
 *
 * SVMipv4::get_score (/root/reference/SVMipv4.cpp:176-246) evaluates
 *     exponent = (-35.0464 - 4) + t1 + t2 + ... + t69          (left-to-right double additions)
 * where each term is  coef*v  |  (coef*v1)*v2  |  coef*(v*v)   in 16 per-candidate variables.
 * The coefficients are the trained model, not code; they are restated here once and consumed by the
 * oracle (same evaluation order as the reference, for bit-exact CPU parity) and by the HIP kernels
 * (which regroup the same terms by sequence window).  A mistranscription is caught by
 * tests/test_oracle_golden.py, which compares the oracle with known answers of the compiled reference (tests/golden/candidates.npz).
 */
#ifndef MIPGEN_LOGISTIC_MODEL_H
#define MIPGEN_LOGISTIC_MODEL_H

/* variable ids */
enum {
    MLV_BPS = 0,   /* bases_per_switch = scan_size / run_count          (SVMipv4.cpp:143) */
    MLV_EGC,       /* ext_gc_content                                     */
    MLV_LGC,       /* lig_gc_content                                     */
    MLV_TLEN,      /* target_length = min(scan_size, 250)                (SVMipv4.cpp:157) */
    MLV_EA,        /* ext_a_content                                      */
    MLV_EG,        /* ext_g_content                                      */
    MLV_ELEN,      /* ext_length                                         */
    MLV_LG,        /* lig_g_content                                      */
    MLV_LEC,       /* log_ext_copy = copy>100 ? 2 : log10(copy)          (SVMipv4.cpp:173) */
    MLV_LLC,       /* log_lig_copy                                       */
    MLV_JS,        /* junction_score                                     (SVMipv4.cpp:171,249-267) */
    MLV_LA,        /* lig_a_content                                      */
    MLV_LLEN,      /* lig_length                                         */
    MLV_TA,        /* target_a_content                                   */
    MLV_TGC,       /* target_gc_content                                  */
    MLV_TG,        /* target_g_content                                   */
    MLV_COUNT
};

enum { MLT_LIN = 0, MLT_BIL = 1, MLT_SQ = 2 };

typedef struct mipgen_logistic_term {
    double coef;
    int kind;      /* MLT_* */
    int v1, v2;    /* v2 unused unless MLT_BIL */
} mipgen_logistic_term;

#define MIPGEN_LOGISTIC_C0 (-35.0464)
#define MIPGEN_LOGISTIC_C1 (4.0)          /* exponent starts as (C0 - C1) */
#define MIPGEN_LOGISTIC_BASE (2.71828)    /* pow(2.71828, x), not exp(x)  (SVMipv4.cpp:247) */
#define MIPGEN_LOGISTIC_NTERMS 69

#define MIPGEN_LOGISTIC_TERMS { \
    {-1.974282, MLT_LIN, MLV_BPS, 0}, \
    {2.63667, MLT_BIL, MLV_BPS, MLV_EGC}, \
    {2.540741, MLT_BIL, MLV_BPS, MLV_LGC}, \
    {-0.006488, MLT_BIL, MLV_BPS, MLV_TLEN}, \
    {-0.018137, MLT_LIN, MLV_EA, 0}, \
    {7.795877, MLT_SQ, MLV_EA, 0}, \
    {-5.576753, MLT_BIL, MLV_EA, MLV_EG}, \
    {-0.274062, MLT_BIL, MLV_EA, MLV_ELEN}, \
    {8.695568, MLT_BIL, MLV_EA, MLV_LG}, \
    {-2.014126, MLT_BIL, MLV_EA, MLV_LEC}, \
    {3.163087, MLT_BIL, MLV_EA, MLV_LLC}, \
    {-19.900678, MLT_SQ, MLV_EGC, 0}, \
    {35.747084, MLT_LIN, MLV_EGC, 0}, \
    {-2.082136, MLT_LIN, MLV_EG, 0}, \
    {7.204324, MLT_SQ, MLV_EG, 0}, \
    {11.73888, MLT_BIL, MLV_EG, MLV_LG}, \
    {-2.173235, MLT_BIL, MLV_EG, MLV_LLC}, \
    {-7.123214, MLT_BIL, MLV_EG, MLV_TGC}, \
    {1.068617, MLT_LIN, MLV_ELEN, 0}, \
    {-0.008666, MLT_SQ, MLV_ELEN, 0}, \
    {-0.555599, MLT_BIL, MLV_ELEN, MLV_EGC}, \
    {-0.289857, MLT_BIL, MLV_ELEN, MLV_LG}, \
    {-0.009621, MLT_BIL, MLV_ELEN, MLV_LLEN}, \
    {0.119863, MLT_BIL, MLV_ELEN, MLV_LEC}, \
    {2.405833, MLT_LIN, MLV_JS, 0}, \
    {-1.764289, MLT_BIL, MLV_JS, MLV_LGC}, \
    {2.112564, MLT_BIL, MLV_JS, MLV_LG}, \
    {0.656183, MLT_BIL, MLV_JS, MLV_LEC}, \
    {-3.099451, MLT_BIL, MLV_JS, MLV_TA}, \
    {-2.097335, MLT_BIL, MLV_JS, MLV_TGC}, \
    {6.542827, MLT_SQ, MLV_LA, 0}, \
    {-8.885956, MLT_LIN, MLV_LA, 0}, \
    {5.297562, MLT_BIL, MLV_LA, MLV_EG}, \
    {13.042436, MLT_BIL, MLV_LA, MLV_LGC}, \
    {-12.333361, MLT_BIL, MLV_LA, MLV_LG}, \
    {19.866468, MLT_LIN, MLV_LGC, 0}, \
    {-13.698276, MLT_SQ, MLV_LGC, 0}, \
    {-0.517301, MLT_BIL, MLV_LGC, MLV_LLEN}, \
    {-2.846777, MLT_BIL, MLV_LGC, MLV_LLC}, \
    {13.64081, MLT_BIL, MLV_LGC, MLV_TA}, \
    {13.614709, MLT_LIN, MLV_LG, 0}, \
    {-4.759165, MLT_BIL, MLV_LG, MLV_EGC}, \
    {-7.48883, MLT_BIL, MLV_LG, MLV_LGC}, \
    {-2.308594, MLT_BIL, MLV_LG, MLV_LEC}, \
    {-12.640154, MLT_BIL, MLV_LG, MLV_TA}, \
    {1.164626, MLT_LIN, MLV_LLEN, 0}, \
    {-0.010326, MLT_SQ, MLV_LLEN, 0}, \
    {-0.354197, MLT_BIL, MLV_LLEN, MLV_EGC}, \
    {0.111448, MLT_BIL, MLV_LLEN, MLV_LEC}, \
    {0.238095, MLT_BIL, MLV_LLEN, MLV_TA}, \
    {-4.161632, MLT_LIN, MLV_LEC, 0}, \
    {-2.728864, MLT_BIL, MLV_LEC, MLV_EGC}, \
    {0.641717, MLT_BIL, MLV_LEC, MLV_LLC}, \
    {3.738798, MLT_BIL, MLV_LEC, MLV_TGC}, \
    {-1.98457, MLT_LIN, MLV_LLC, 0}, \
    {2.362253, MLT_BIL, MLV_LLC, MLV_EGC}, \
    {3.467229, MLT_BIL, MLV_LLC, MLV_TGC}, \
    {-18.443242, MLT_LIN, MLV_TA, 0}, \
    {20.89245, MLT_SQ, MLV_TA, 0}, \
    {-0.048679, MLT_BIL, MLV_TA, MLV_TLEN}, \
    {-50.249451, MLT_SQ, MLV_TGC, 0}, \
    {27.132716, MLT_LIN, MLV_TGC, 0}, \
    {-0.050633, MLT_BIL, MLV_TGC, MLV_TLEN}, \
    {-20.772366, MLT_LIN, MLV_TG, 0}, \
    {-60.796481, MLT_SQ, MLV_TG, 0}, \
    {26.630245, MLT_BIL, MLV_TG, MLV_TA}, \
    {87.162648, MLT_BIL, MLV_TG, MLV_TGC}, \
    {0.030256, MLT_BIL, MLV_TG, MLV_TLEN}, \
    {0.032811, MLT_LIN, MLV_TLEN, 0} }

/* junction_scores (SVMipv4.cpp:249-267), indexed by 4*code(b0)+code(b1) with A=0,C=1,G=2,T=3;
 * any junction containing another character scores 0.0 (std::map::operator[] default). */
#define MIPGEN_JUNCTION_SCORES { \
    0.0, 0.35, 0.046, 0.079, \
    0.34, 0.22, 0.55, -0.071, \
    0.35, 0.92, 0.24, 0.48, \
    -0.46, -0.35, -0.25, -0.98 }

/* the 44 long-range mers (mipgen.cpp:32), in feature order 22..65 */
#define MIPGEN_FEATURE_MERS { \
    "A","AA","AAA","AAC","AAG","AAT","AC","ACA","ACC","ACG","AG","AGA","AGC","AGG","AGT", \
    "AT","ATA","ATC","ATG","CAG","CG","CGG","G","GAC","GAG","GC","GCG","GG","GGC","GGG","GTG", \
    "TA","TAA","TAC","TAG","TC","TCC","TCG","TG","TGA","TGC","TGG","TTC","TTG" }

#endif
