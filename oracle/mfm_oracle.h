/*
 * mfm_oracle.h - CPU restatement of the multifm channel hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X multifm engine.  It is test infrastructure:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (tsl-sdr_amd/) never links, imports or calls anything in oracle/.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - fast_atan2f ............ PINNED.  oracle/_ref/libref_fast_atan2f.so is built from the
 *                              reference's own multifm/fast_atan2f.c (it needs no header the
 *                              image lacks); tests compare mfmo_fast_atan2f against it on dense
 *                              sweeps, and tests/golden/atan2_golden.npz holds vectors made by it.
 *   - taps / rotator / FIR / derotation / discriminator glue ... PARITY UNPINNED.  These
 *                              reference files include <tsl/...> headers that are not in the
 *                              image, so they are unbuildable here; the reference's own tests
 *                              hold no golden vectors for them (filter/test/test_direct_fir.c:19-24
 *                              is empty).  They are restated from the cited lines below.
 *
 * All citations are relative to the reference tree (pvachon/tsl-sdr).
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* filter/filter.h:16 - "Q15" is really Q14. */
#define MFMO_Q_SHIFT 14

/* ---- scalar arithmetic pieces ------------------------------------------------------- */

/* filter/complex.h:30-34 (round_q30_q15): (a >> 14) + ((a >> 13) & 1), truncated to int16. */
int16_t mfmo_r14(int32_t a);

/* multifm/fast_atan2f.c:14-81: 257-entry table, entry i = atan(min(i,255)/255) printed with 7
 * significant digits ("%.6e") and read back as a float.  Generated, not transcribed. */
void mfmo_atan_table(float tbl[257]);

/* multifm/fast_atan2f.c:101-174, no FP contraction (canonical: the reference's default CMake
 * configuration has no -O flag, so gcc never fuses line 131). */
float mfmo_fast_atan2f(float y, float x);
/* Same with line 131 fused into one fma (what -O2 -march=native -ffp-contract=fast yields on
 * an FMA host); used only to bound the 1-LSB ambiguity. */
float mfmo_fast_atan2f_fma(float y, float x);

/* multifm/fm_demod.c:71-72: (int16)(float)(((double)phi / M_PI) * 16384.0). */
int16_t mfmo_phi_to_pcm(float phi);

/* out[i] = mfmo_phi_to_pcm(float with bit pattern first_bits + i); exhaustive-scan helper */
void mfmo_phi_to_pcm_range(uint32_t first_bits, uint32_t count, int16_t *out);

/* out[i] = pcm of fast_atan2f((float)s_im[i], (float)s_re[i]) (fm_demod.c:68-72); fused selects the
 * fma variant of fast_atan2f.c:131 */
void mfmo_discriminate_batch(const int32_t *s_re, const int32_t *s_im, size_t n, int16_t *out, int fused);

/* One discriminator step, multifm/fm_demod.c:55-72.  prev/cur are (re,im) int16 pairs. */
int16_t mfmo_fm_step(int16_t a_re, int16_t a_im, int32_t last_re, int32_t last_im);

/* ---- per-channel set-up ------------------------------------------------------------- */

/* multifm/demod.c:210,232-243 (_demod_fir_prepare): rotated Q14 taps, truncated toward zero. */
void mfmo_make_taps(const double *lpf_taps, size_t nr_taps, int32_t offset_hz, uint32_t sample_rate,
                    double gain, int16_t *coeff_re, int16_t *coeff_im);

/* filter/direct_fir.c:72-79 (direct_fir_init): rotator increment. */
void mfmo_rot_incr(int32_t offset_hz, uint32_t sample_rate, unsigned decimation, int16_t *incr_re,
                   int16_t *incr_im);

/* multifm/receiver.c:218-220: linear gain = 10^(dB/10). */
double mfmo_gain_from_db(double db);

/* One rotator step, filter/direct_fir.c:151-172 + filter/complex.h:51-62. */
void mfmo_rot_step(int16_t *rot_re, int16_t *rot_im, int16_t incr_re, int16_t incr_im);

/* ---- streaming channel -------------------------------------------------------------- */

struct mfmo_chan;

/* A channel = direct_fir (filter/direct_fir.h:9-74) + fm demod state (multifm/fm_demod.c:14-18). */
struct mfmo_chan *mfmo_chan_new(const int16_t *coeff_re, const int16_t *coeff_im, size_t nr_taps,
                                unsigned decimation, int16_t incr_re, int16_t incr_im);
void mfmo_chan_free(struct mfmo_chan *ch);

/*
 * Feed nr_samples interleaved int16 IQ samples (any chunking).  Appends up to max_out outputs:
 * pcm_out[n] (int16 PCM, multifm/demod.c:89-93 FIFO stream) and, if iq_out != NULL,
 * iq_out[2n],iq_out[2n+1] (the signalDebugFile stream, multifm/demod.c:75-81).
 * Returns the number of outputs produced.  Stream semantics: output n uses samples
 * [n*D, n*D+T) (filter/direct_fir.c:343-391); first output at n = 0; no zero history.
 */
size_t mfmo_chan_feed(struct mfmo_chan *ch, const int16_t *iq, size_t nr_samples, int16_t *pcm_out,
                      int16_t *iq_out, size_t max_out);

/* Advance a channel's rotator as if nr_outputs outputs had been produced (filter/direct_fir.c:166-167 that many times):
 * lets a test check a window in the middle of a long stream.  Feed the window one output early and drop that output's PCM
 * (the discriminator's previous sample is not reconstructed). */
void mfmo_chan_skip_outputs(struct mfmo_chan *ch, uint64_t nr_outputs);

/* Rotator state after the outputs produced so far (direct_fir.h:52-64). */
void mfmo_chan_rot(const struct mfmo_chan *ch, int16_t *rot_re, int16_t *rot_im);

/*
 * Whole-buffer helper used by the tests and by bench.py's cpu_baseline: run nr_chan channels over
 * one contiguous IQ buffer with nr_threads worker threads, channels dealt round-robin to threads
 * (the reference runs one thread per channel, multifm/receiver.c:89-95).  coeff_* are
 * [nr_chan][nr_taps], incr is [nr_chan][2], pcm_out is [nr_chan][out_stride].
 * Returns outputs per channel.
 */
size_t mfmo_run_channels(const int16_t *iq, size_t nr_samples, size_t nr_chan, const int16_t *coeff_re,
                         const int16_t *coeff_im, size_t nr_taps, unsigned decimation,
                         const int16_t *incr, int16_t *pcm_out, int16_t *iq_out, size_t out_stride,
                         unsigned nr_threads);

/*
 * Structure-following variant of the reference's two-slot buffer walk
 * (filter/direct_fir.c:118-146 push, :328-417 process, :455-472 can_process and the driver loop
 * multifm/demod.c:58-115), used to show the closed-form stream above is what that walk yields
 * for uniform buffers.  Feeds `nr_bufs` buffers of `buf_samples` samples taken back to back
 * from iq.  Returns outputs produced.
 */
size_t mfmo_twoslot_run(const int16_t *iq, size_t buf_samples, size_t nr_bufs, const int16_t *coeff_re,
                        const int16_t *coeff_im, size_t nr_taps, unsigned decimation, int16_t incr_re,
                        int16_t incr_im, int16_t *pcm_out, int16_t *iq_out, size_t max_out);

/* ---- PCM stage behind the FIFO: rational resampler + DC blocker ---------- */

struct mfmo_resampler;

/*
 * filter/polyphase_fir.c:47-105: taps (Q14 int16, decoder/decoder.c:530-533) scattered into `interpolate`
 * phase filters, tap i -> phase i % I, slot i / I, phase length rounded up to a multiple of 4 (zero filled).
 */
struct mfmo_resampler *mfmo_resampler_new(const int16_t *coeffs, size_t nr_coeffs, unsigned interpolate,
                                          unsigned decimate);
void mfmo_resampler_free(struct mfmo_resampler *r);
size_t mfmo_resampler_phase_len(const struct mfmo_resampler *r);

/*
 * Feed real int16 PCM (any chunking), append up to max_out outputs.  Stream semantics of
 * polyphase_fir_process (filter/polyphase_fir.c:162-233) + dot_product_sample_buffers_real
 * (filter/utils.c:46-116): output m = r14(sum_k phase[p_m][k] * x[pos_m + k]), int32 wrap; then
 * p += D, pos += p / I, p %= I (:206-211).  An output is produced only while MORE than phase_len
 * unconsumed samples exist (the strict '>' at :184).  Returns outputs produced.
 */
size_t mfmo_resampler_feed(struct mfmo_resampler *r, const int16_t *pcm, size_t nr_samples, int16_t *out,
                           size_t max_out);

/* filter/dc_blocker.h:45-93: differentiator + leaky integrator, sequential, in place. */
struct mfmo_dc_blocker {
    int16_t p;
    int32_t x_n_1, y_n_1, acc;
};
void mfmo_dc_blocker_init(struct mfmo_dc_blocker *b, double pole);
void mfmo_dc_blocker_apply(struct mfmo_dc_blocker *b, int16_t *samples, size_t nr_samples);

/* decoder/decoder.c:530-533: double taps -> Q14 int16 by truncation */
void mfmo_resampler_quantize_taps(const double *taps, size_t n, int16_t *out);

/* ---- 8-bit ingest -------------------------------------------------------------
 * format 1: multifm/file_if.c:66-111 (cs8: sign extension); 2: multifm/file_if.c:113-157 (cu8: bytes read as signed,
 * minus 127, the remainder loop of :146-150 storing the bare cast); 3: multifm/rtl_sdr_if.c:146-158 ((u8 - 127) << 7).
 * nr_bytes = bytes of ONE read (2 per sample); out receives nr_bytes int16 values. */
void mfmo_unpack_bytes(const uint8_t *in, size_t nr_bytes, int format, int16_t *out);

#ifdef __cplusplus
}
#endif
