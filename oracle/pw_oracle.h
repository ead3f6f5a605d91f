/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement ("port") of the reference's PW_ReAligner hot path, used only as the checker:
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * repeatresolver_amd/ may.  Pinned against the compiled reference (oracle/_ref/pw_ref) through the
 * fixtures in tests/golden/ (see oracle/gen_golden.py and tests/test_oracle_golden.py).
 *
 * Every function cites the PW_ReAligner.c lines ("PW:") it restates.
 */
#ifndef PW_ORACLE_H
#define PW_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pwo_state pwo_state;

/* PW:93-241 MMA_Einlesen on an in-memory character matrix (T rows of W chars, row-major, no
 * newlines).  Returns NULL on out-of-memory. */
pwo_state *pwo_create(int T, int W, const unsigned char *rows, int bandwidth);
/* Reads the text file like PW:118-136 (every line must end in '\n', equal lengths required).
 * Returns NULL and writes a message to err (may be NULL) on failure. */
pwo_state *pwo_load(const char *path, int bandwidth, char *err, int errcap);
void pwo_destroy(pwo_state *s);

int pwo_rows(const pwo_state *s);
int pwo_width(const pwo_state *s);               /* current Breite */
int pwo_row_length(const pwo_state *s, int k);   /* Lengths[k] */
uint64_t pwo_cells(const pwo_state *s);          /* DP cells filled so far (PW:1503-1510 executions) */

void pwo_trim(pwo_state *s);                     /* PW:459-645 EntAlGapper */
void pwo_compact(pwo_state *s);                  /* PW:706-763 W_Con */
int pwo_check_tallies(const pwo_state *s);       /* PW:765-859 W_Con_Checker: 0 = consistent */
int pwo_realign_row(pwo_state *s, int k);        /* PW:1469-1531 Matrix_Filler(k): 0 ok, <0 error */
void pwo_realign_round(pwo_state *s);            /* PW:1695-1737 */
uint64_t pwo_total_score(pwo_state *s);          /* PW:933-963 (compacts first, like the reference) */
void pwo_export(const pwo_state *s, unsigned char *out); /* PW:1556-1598: T*W chars "ACGT- ", no newlines */

/* Whole program, PW:1610-1759: same stdout lines (except the CPU-time line), same file writes.
 * Returns the exit code the reference would use. max_rounds < 0 = unlimited (10000 like PW:1681). */
int pwo_run(const char *in_path, const char *out_path, int bandwidth, FILE *log, int max_rounds);

/* column ordinals (current numbering; call pwo_compact first for the compacted one) of the bases of row k, left to
 * right; returns their number (at most cap are written) */
int pwo_row_columns(const pwo_state *s, int k, int *out, int cap);

/* ---- introspection for kernel-level parity tests (state of the LAST pwo_realign_row call) ---- */
int pwo_dbg_L(const pwo_state *s);
int pwo_dbg_W_at_fill(const pwo_state *s);
const int *pwo_dbg_way(const pwo_state *s);              /* Way[x], PW:647-705 */
const unsigned char *pwo_dbg_seq(const pwo_state *s);    /* Seq_Bases[x] */
/* tallies with row k removed, as the fill saw them: W_at_fill x 6 values */
const uint64_t *pwo_dbg_tallies(const pwo_state *s);
/* new placement: for base x, column ordinal (pre-insertion numbering) and 1 if a new column was
 * opened after that ordinal (PW:1404-1410), else 0 */
const int *pwo_dbg_newcol(const pwo_state *s);
const unsigned char *pwo_dbg_newins(const pwo_state *s);
int pwo_dbg_entry(const pwo_state *s);                   /* "wayin", PW:1352-1360 */
/* band matrix value M[x][j] (j = y - anf(x)); UINT64_MAX/2-ish values are INF */
uint64_t pwo_dbg_M(const pwo_state *s, int x, int j);
/* dev aids (scripts/dev/start_vectors.py): the fill of row k against the state as it is, the state left unchanged; one
 * whole DP row of the last fill (B values, band-relative) */
int pwo_fill_only(pwo_state *s, int k);
const uint64_t *pwo_dbg_Mrow(const pwo_state *s, int x);

#ifdef __cplusplus
}
#endif
#endif
