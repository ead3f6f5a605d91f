/*
 * pwr.h -- C ABI of the MI355X-native PW_ReAligner hot path (libpwr.so).
 *
 * The reference (PhilippBongartz/RepeatResolver, PW_ReAligner.c, cited as "PW:") has no library
 * interface: its boundary is the process (`./PW_ReAligner <MSA> [-o out] [-b bw]`) and, inside it,
 * a handful of functions working on file-scope globals.  Each entry point below replaces one of
 * those functions; the context object replaces the globals (PW:30-52, PW:86-90).  Plain C types
 * only, `int` status returns (0 = ok, negative = error), no callbacks, a context is not
 * thread-safe.  The MSA state lives in HBM for the lifetime of the context.
 *
 * Symbols: the text alphabet is `acgtACGT-_` and ' ' on input (PW:165-222) and `ACGT- ` on
 * output (PW:1558-1563).
 */
#ifndef PWR_H
#define PWR_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWR_OK 0
#define PWR_ERR_ARG (-1)         /* bad argument */
#define PWR_ERR_NOMEM (-2)       /* host or device allocation failed (PW:55-75) */
#define PWR_ERR_DEVICE (-3)      /* HIP runtime error / no usable GPU */
#define PWR_ERR_INPUT (-4)       /* malformed MSA text (PW:121, PW:134; unequal line lengths) */
#define PWR_ERR_RANGE (-5)       /* a limit was exceeded (row longer than 35000 bases PW:16,
                                    bandwidth > 2000 PW:14).  Scores beyond the 32-bit range of the fast
                                    fill are NOT an error: such rows take the 64-bit fill (pwr_stats.rows_wide) */
#define PWR_ERR_INTERNAL (-6)    /* inconsistent traceback ("Stuff gone wrong", PW:1412-1427) */
#define PWR_ERR_UNSUPPORTED (-7) /* untrimmed state: call pwr_trim_ends first (see pwr_create) */
#define PWR_ERR_IO (-8)          /* output file cannot be opened ("DateiVerbratei!", PW:1568-1572) */
#define PWR_ERR_STALL (-9)       /* a wave of the fill kernel timed out waiting for a neighbour work-group (GPU shared or
                                    oversubscribed) AND the geometry has no one-work-group form to repeat the job with
                                    ("waves" = 17); with the other geometries a stall only costs time (pwr_stats.stalls) */
#define PWR_ERR_ORDER (-10)      /* a row that was realigned ahead of rows outside its batch ("plan_ahead") was found, at the gather of
                                    one of those rows, to lie within reach of it after all: the two do not commute and the state is
                                    not the reference's.  Never reached with the gap and the event rate rows jump at; the check
                                    exists so that it could not go unnoticed.  Rerun with "plan_ahead" 0. */

#define PWR_MAX_BANDWIDTH 2000   /* PW:14 */
#define PWR_MAX_SEQ_LENGTH 35000 /* PW:16 */
#define PWR_MAX_LINE 699997      /* PW:15, PW:119: fgets buffer of Max_MA_Breadth-2 */

typedef struct pwr_ctx pwr_ctx;

typedef struct pwr_stats {
    uint64_t cells_reference;   /* DP cells the reference would have filled for the committed
                                   realignments (one execution of PW:1503-1510 each) */
    uint64_t cells_computed;    /* DP cells the fill kernel actually computed (includes speculative
                                   fills that had to be repeated) */
    uint64_t fill_launches;     /* launches of the DP fill kernel */
    double fill_ms;             /* sum of HIP-event durations of the timed launches (profiling on) */
    uint64_t rows_committed;    /* realignments committed (== Matrix_Filler calls with length>0) */
    uint64_t rows_recomputed;   /* speculative results discarded because their inputs changed */
    uint64_t batches;           /* speculative batches launched */
    uint64_t rows_changed;      /* committed realignments that changed the MSA */
    uint64_t reject_reason[4];  /* speculative rejections: interval ends/length, left clamp, right clamp, newer column */
    uint64_t fill_launches_timed; /* launches covered by fill_ms (the first 65536 after a reset) */
    uint64_t stalls;            /* k_fill_v3 jobs given up after a time-out and repeated by k_fill_v2 */
    uint64_t rows_ahead;        /* commits that went ahead of a stale row of their batch (the two rows commute: disjoint band intervals) */
    uint64_t rows_wide;         /* committed realignments whose scores were not provably below 2^30 and that the 64-bit fill
                                   (the reference's own arithmetic, PW:30, PW:271) computed */
    uint64_t seg_jobs;          /* fills that ran as several segments side by side (k_fill_v3) */
    uint64_t segs;              /* ... and the segments they were cut into */
    uint64_t seg_fails;         /* ... of which this many failed the check of a segment's start and were repeated in one piece */
    uint64_t rows_jumped;       /* commits of rows picked ahead of rows that were not in their batch ("plan_ahead") */
} pwr_stats;

/* Replaces MMA_Einlesen (PW:93-241) for an in-memory matrix: `text` holds rows*width characters,
 * row-major, without newlines.  device = HIP device ordinal.  The rows are parsed and kept on the
 * host until the first device operation.  Realignment needs every run of non-blank cells of a row to
 * begin and end with a base, which is what pwr_trim_ends() (EntAlGapper) establishes for ANY input and
 * what every reference-written MSAreal satisfies -- the reference itself always trims first (PW:1655).
 * Rows with blank runs between their bases are fine (they are kept as chains of segments until their
 * first realignment joins them, exactly as PW:1362-1443 does).  A state that was not trimmed ('-' next to
 * a blank or at an MSA edge) makes the device calls return PWR_ERR_UNSUPPORTED. */
int pwr_create(pwr_ctx **out, int rows, int width, const unsigned char *text, int bandwidth, int device);
void pwr_destroy(pwr_ctx *ctx);

/* EntAlGapper (PW:459-645). */
int pwr_trim_ends(pwr_ctx *ctx);
/* Matrix_Filler(k) (PW:1469-1531) including W_Con (PW:706-763) before it. */
int pwr_realign_row(pwr_ctx *ctx, int k);
/* The k-loop of one round (PW:1695-1737): every row, in input order.  Results are those of
 * calling pwr_realign_row for k = 0..rows-1; internally upcoming rows are filled speculatively and
 * committed in order. */
int pwr_realign_round(pwr_ctx *ctx);
/* A slab of that k-loop: rows k0 .. k0+n-1 in input order (a partial PW:1695 loop).  Calling it for consecutive
 * slabs that cover 0..rows-1 equals pwr_realign_round; speculation stays inside the slab. */
int pwr_realign_rows(pwr_ctx *ctx, int k0, int n);
/* The same slab with the work of every speculative batch split over `world` contexts, one per GPU of a node, each a
 * replica built from the same text with the same options (SURVEY 8e "within one MSA": the rows of the k-loop window PW:1695
 * are realigned side by side against the last committed state and committed in the reference's order; a replica fills and
 * traces the jobs j of a batch with j % world == rank, and commits them all).  Per batch the caller all-gathers one record
 * per job (RCCL over xGMI when the ranks are GPUs of a node; this library links no communication library):
 *     pwr_split_begin(ctx, k0, n, rank, world);
 *     do { pwr_split_stage(ctx, send); <all-gather send -> recv>; pwr_split_commit(ctx, recv, &left); } while (left > 0);
 * send: slots_per_rank * slot_bytes of DEVICE memory (pwr_split_slot_bytes), recv: world times that, the ranks' parts in
 * rank order.  "window" (set before the first device call) is the batch size, world <= window.  The MSA afterwards is the
 * one pwr_realign_rows(k0, n) leaves on one GPU, on every replica. */
int pwr_split_begin(pwr_ctx *ctx, int k0, int n, int rank, int world);
int pwr_split_slot_bytes(pwr_ctx *ctx, size_t *slot_bytes, int *slots_per_rank);
int pwr_split_stage(pwr_ctx *ctx, void *send_dev);
int pwr_split_commit(pwr_ctx *ctx, const void *recv_dev, int *rows_left);
/* The integer total that OverallScorePrint prints (PW:864-892, PW:933-963); compacts first. */
int pwr_total_score(pwr_ctx *ctx, uint64_t *total);
/* Tiefe / current Breite (PW:86-87). */
int pwr_dims(pwr_ctx *ctx, int *rows, int *width);
/* MMA_Auslesen (PW:1556-1598) into memory: rows*width characters, row-major, no newlines. */
int pwr_export_rows(pwr_ctx *ctx, unsigned char *buf, size_t cap);
/* MMA_Auslesen (PW:1556-1598) without waiting for it.  pwr_snapshot_begin takes the image of the FILE the reference would
 * write now -- `rows` lines of `width` characters "ACGT- ", each followed by '\n' -- on the device, in stream order (calls
 * that follow may change the state at once), and starts copying it to page-locked host memory on a stream of its own;
 * pwr_snapshot_wait (any thread) blocks until the image is there: *image stays valid until pwr_snapshot_free.  One snapshot
 * per context at a time.  pwr_run_file hands the image to a writer thread, so the rewrite of the output after every
 * improving round (PW:1741) does not hold up the next round. */
typedef struct pwr_snapshot pwr_snapshot;
int pwr_snapshot_begin(pwr_ctx *ctx, pwr_snapshot **snap);
int pwr_snapshot_wait(pwr_snapshot *snap, const unsigned char **image, size_t *bytes, int *rows, int *width);
void pwr_snapshot_free(pwr_snapshot *snap);

/* Knobs and counters (ours).  Results never depend on any of them.  keys:
 *   "window"    rows gathered / filled per batch, the first certain to commit, the others speculative (1..128, default 3)
 *   "profile"   1 = time every fill launch with HIP events (pwr_stats.fill_ms)
 *   "fill"      DP fill kernel: 4 = k_fill_v3 (default: one work-group per pipeline wave, the DP in segments side by side);
 *               3 = k_fill_v2 (one work-group per DP, in one piece: the independent cross-check and the stand-in after a stall)
 *   "waves"     waves per DP segment of the wave-pipeline fills: 5 (default, 4 columns per lane), 9, 8, 4, 3, or 17 with
 *               k_fill_v3 only; bandwidths above 1000 always use 9
 *   "onewg"     1 = the waves of a k_fill_v3 segment form ONE work-group and hand over through LDS (default 0: measured slower)
 *   "seg_rows", "seg_max", "warm_pct", "seg_align", "src_start"
 *               k_fill_v3 fills a DP as up to seg_max (<= 256, default 64) segments of about seg_rows (default 160) rows side
 *               by side, each warmed up while the band moves by warm_pct (default 190) percent of the bandwidth -- from the
 *               column of the base before its first row alone (src_start 1, default) or from the free start of PW:265 (0); with
 *               "warm_adapt" 1 (default, needs src_start) warm_pct is the upper bound of a length that follows the failures of the
 *               check down to "warm_min_pct" (100): "warm_down_pm" (5) per mille of the bandwidth off per fill that passes,
 *               "warm_up_pm" (50) back on per fill that fails; read-only "warm_now" = the present length in percent -- and CHECKS
 *               every segment's start; a row whose check fails is repeated with twice the warm-up, then in
 *               one piece (pwr_stats.seg_fails).  seg_rows 0 = always in one piece; seg_align (16, 32, 64): the segments' first rows are
 *               multiples of it
 *   "seg_budget", "seg_minrows"
 *               seg_budget > 0: the segments the jobs of one batch may have TOGETHER, dealt to them by their rows' lengths -- a long
 *               row beside two short ones is cut finer than one of three long rows --, none with fewer own rows than seg_minrows
 *               (default 64), none with more segments than seg_max; default 0 = about seg_rows rows per segment whatever the
 *               batch (a budget of 160-320 was measured 7-20 % slower: every further segment brings its own warm-up)
 *   "plan_ahead"
 *               1 (default): the speculative rows of a batch are picked by the batch before it -- first the rows among the next 64
 *               whose band interval keeps more than 2048 columns from that of every uncommitted row before them (they commute with
 *               all of those, so they commit in this very batch), then the next rows in order as before; rows are picked ahead only
 *               while a commit opens / empties fewer than 12 columns on average, and every row that was jumped checks at its gather
 *               that the gap has held (PWR_ERR_ORDER otherwise); 0: the next rows in order (round 3).  "window" above 16: in order.
 *               ("plan_slack", "plan_evrate_x100": the gap and the rate, test hooks; with a smaller gap rows also stop jumping as soon as 64
 *               commits at the present rate would fill half of it, "plan_gate_rel" 1.)
 *   "hard_rows", "hard_up_pm", "hard_down_pm"
 *               hard_rows 1 (default): a row whose segment check fails warms up over hard_up_pm (300) per mille of the bandwidth more
 *               than the steered length in all its later fills, hard_down_pm (0) less again after each of its commits -- failures
 *               are a property of the row (a row that failed once fails again in 46 % of its fills, any row in 8 %: DESIGN.md
 *               3.2), so the rows that need a long warm-up get one and the steered length of all others comes down; 0: one
 *               length for all (round 3); 2: marked and counted only (read-only "hard_marked", "hard_fills", "hard_refail")
 *   "check_in_trace"
 *               1 (default): the work-groups that check the segments' starts ride in the traceback kernel's launch (k_trace_blk) instead of
 *               a launch of their own between fill and traceback; 0: their own launch
 *   "fail_stops"
 *               1 (default): a job whose segment check failed ends its batch (round 3's rule); 0 (experimental): it waits for its
 *               repeat like a stale row -- later rows of the batch whose band intervals are disjoint from its own commit ahead of
 *               it; 0.7 % faster, but one case of the randomised sweep fails with it: not for production
 *   "seg_balance"
 *               1: the own parts of a job's segments are cut so that every segment runs about the same number of rows, its warm-up
 *               included (the first has none; a warm-up of so many columns is more rows where the bases sit closer); default 0 =
 *               equal own parts (the balanced cut was measured 5-12 % slower: the first segment's long own part, whose rows make
 *               the record, becomes the launch's last)
 *   "ptrace"    traceback kernel: 2 = k_trace_blk (default: one wave per 64 rows, no hand-over chain), 1 = k_trace_par (64 chunks
 *               handing over top-down), 0 = k_trace_wp (one wave per job)
 *   "slack"     spare column capacity kept when the device arrays are (re)allocated
 *   "spec_len"  percent a speculative row may be longer than the first row of its batch (default 6)
 * "fill", "waves", "slack", "src_start" and the "seg" options must be set before the first call that touches the device. */
int pwr_set_option(pwr_ctx *ctx, const char *key, long value);
int pwr_get_option(pwr_ctx *ctx, const char *key, long *value);
int pwr_get_stats(pwr_ctx *ctx, pwr_stats *out);
int pwr_reset_stats(pwr_ctx *ctx);
const char *pwr_strerror(int code);
/* Number of HIP devices visible, or a negative error. */
int pwr_device_count(void);

/* ---- host side of the boundary, plain C (pwr_host.c) ---- */
/* Reads the MSA text file exactly as PW:118-136 does, but refuses unequal line lengths. On
 * success *text is malloc'ed (rows*width bytes). */
int pwr_read_msa_file(const char *path, int *rows, int *width, unsigned char **text, char *err, size_t errcap);
int pwr_write_msa_file(const char *path, int rows, int width, const unsigned char *text);
/* main() of the reference (PW:1610-1759) on top of the calls above: same stdout lines, same
 * output-file behaviour, returns the process exit code. max_rounds < 0: until convergence. */
int pwr_run_file(const char *in_path, const char *out_path, int bandwidth, int device, int max_rounds, FILE *log);

#ifdef __cplusplus
}
#endif
#endif /* PWR_H */
