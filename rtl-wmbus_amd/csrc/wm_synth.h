/* wm_synth.h -- synthetic cu8 capture generator (see wm_synth.c). */
#ifndef WM_SYNTH_H
#define WM_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { WMSYNTH_T1 = 1, WMSYNTH_C1A = 2, WMSYNTH_C1B = 4, WMSYNTH_S1 = 8 };

typedef struct wmsynth_cfg {
    uint64_t seed;
    unsigned fs_khz;            /* d * 800 */
    double noise_sigma;         /* LSB per component */
    double amplitude;           /* LSB */
    double frames_per_s;        /* Poisson rate of bursts; 0 = noise only */
    unsigned kinds;             /* bitmask of WMSYNTH_* */
    int l_min, l_max;           /* L-field range (CRC-free byte count after L) */
    double t1c1_center_khz;     /* +325 for the -s layout, else 0 */
    double s1_center_khz;       /* -325 for the -s layout, else 0 */
    double max_offset_khz;      /* carrier offset drawn from U(-max, +max) */
} wmsynth_cfg;

typedef struct wmsynth_frame {
    uint32_t kind;              /* WMSYNTH_*                                             */
    uint32_t n_samples;         /* burst length in input samples                         */
    uint64_t start_sample;
    uint16_t len;               /* bytes in telegram[]                                   */
    uint8_t  complete;          /* burst fits entirely in the capture                    */
    uint8_t  pad;
    uint8_t  telegram[256];     /* CRC-free telegram = the hex payload the receiver prints */
} wmsynth_frame;

void   wmsynth_default_cfg(wmsynth_cfg *cfg);
/* Fills out[0 .. 2*n_samples) and returns the number of bursts placed (frames[] receives the
 * first frames_cap of them; may be NULL). */
size_t wmsynth_generate(const wmsynth_cfg *cfg, uint8_t *out, size_t n_samples,
                        wmsynth_frame *frames, size_t frames_cap);
#ifdef __cplusplus
}
#endif
#endif
