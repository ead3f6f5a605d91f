/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See pw_oracle.h.
 *
 * CPU restatement of /root/reference/PW_ReAligner.c ("PW").  The MSA is an ordered array of
 * column slots (the reference uses a doubly linked list, PW:41-52); everything observable --
 * scores, tie-breaks, written bytes -- follows the reference line by line as cited.
 * Symbols: 0..3 = A,C,G,T; 4 = '-'; 5 = ' ' (PW:165-222).
 * Tallies w[b] = number of rows in the column that are non-blank and != b (PW:170-217).
 */
#define _POSIX_C_SOURCE 200809L
#include "pw_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PWO_INF (UINT64_MAX / 2)          /* PW:271 Max_Long/2 */
#define MAX_LINE 700000                   /* PW:15 Max_MA_Breadth */
#define MAX_SEQ_LEN 35000                 /* PW:16 */
#define MAX_BAND 2000                     /* PW:14 */

struct pwo_state {
    int T, W, B, H;
    int cap;                  /* allocated slots */
    int nslots;               /* slots ever handed out */
    unsigned char *sym;       /* [slot][T] */
    uint64_t *w;              /* [slot][6] */
    int *order;               /* ordinal -> slot */
    int *scratch_order;
    int *freel;
    int nfree;
    int *lengths;
    uint64_t cells;
    /* per-realignment scratch (PW:30-35) */
    uint64_t *M;
    size_t Mcap;
    uint64_t *G;              /* G[y+1] = sum_{j<=y} S(j,4), for the virtual extension PW:285-295 */
    int *way;
    unsigned char *seqb;
    int L;
    int Wfill;
    uint64_t *dbg_tallies;
    int *newcol;
    unsigned char *newins;
    int entry;
    int *pend_after, *pend_slot;
    int npend;
};

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); return p; }

static int ensure_slots(pwo_state *s, int need)
{
    if (need <= s->cap) return 0;
    int ncap = s->cap ? s->cap : 16;
    while (ncap < need) ncap += ncap / 2 + 16;
    unsigned char *ns = realloc(s->sym, (size_t)ncap * (size_t)s->T);
    if (!ns) return -1;
    s->sym = ns;
    uint64_t *nw = realloc(s->w, (size_t)ncap * 6 * sizeof(uint64_t));
    if (!nw) return -1;
    s->w = nw;
    int *p;
    p = realloc(s->order, (size_t)ncap * sizeof(int));
    if (!p) return -1;
    s->order = p;
    p = realloc(s->scratch_order, (size_t)ncap * sizeof(int));
    if (!p) return -1;
    s->scratch_order = p;
    p = realloc(s->freel, (size_t)ncap * sizeof(int));
    if (!p) return -1;
    s->freel = p;
    p = realloc(s->pend_after, (size_t)ncap * sizeof(int));
    if (!p) return -1;
    s->pend_after = p;
    p = realloc(s->pend_slot, (size_t)ncap * sizeof(int));
    if (!p) return -1;
    s->pend_slot = p;
    uint64_t *g;
    g = realloc(s->G, ((size_t)ncap + 2) * sizeof(uint64_t));
    if (!g) return -1;
    s->G = g;
    g = realloc(s->dbg_tallies, (size_t)ncap * 6 * sizeof(uint64_t));
    if (!g) return -1;
    s->dbg_tallies = g;
    s->cap = ncap;
    return 0;
}

static int new_slot(pwo_state *s)
{
    if (s->nfree > 0) return s->freel[--s->nfree];      /* PW:1261-1265 take from Reservoir */
    if (ensure_slots(s, s->nslots + 1)) return -1;
    return s->nslots++;
}

static inline unsigned char *col_sym(const pwo_state *s, int slot) { return s->sym + (size_t)slot * s->T; }
static inline uint64_t *col_w(const pwo_state *s, int slot) { return s->w + (size_t)slot * 6; }

/* PW:165-222: add one symbol to a column's tallies */
static inline void tally_add(uint64_t *w, int symb)
{
    if (symb == 5) return;
    for (int b = 0; b < 6; b++) if (b != symb) w[b] += 1;
}
static inline void tally_sub(uint64_t *w, int symb)
{
    if (symb == 5) return;
    for (int b = 0; b < 6; b++) if (b != symb) w[b] -= 1;
}

static int code_of(unsigned char c)
{
    switch (c) {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    case '-': case '_': return 4;
    case ' ': return 5;
    default: return -1;   /* the reference leaves the cell uninitialised (PW:165-222 has no else) */
    }
}

pwo_state *pwo_create(int T, int W, const unsigned char *rows, int bandwidth)
{
    if (T <= 0 || W <= 0 || bandwidth < 1 || bandwidth > MAX_BAND) return NULL;
    pwo_state *s = calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->T = T; s->W = W; s->B = bandwidth; s->H = bandwidth / 2;   /* PW:1625-1626, 1639-1640 */
    if (ensure_slots(s, W + W / 10 + 16)) { pwo_destroy(s); return NULL; }
    s->nslots = W;
    s->lengths = calloc((size_t)T, sizeof(int));
    s->way = xmalloc(sizeof(int) * (MAX_SEQ_LEN + 1));
    s->seqb = xmalloc(MAX_SEQ_LEN + 1);
    s->newcol = xmalloc(sizeof(int) * (MAX_SEQ_LEN + 1));
    s->newins = xmalloc(MAX_SEQ_LEN + 1);
    if (!s->lengths || !s->way || !s->seqb || !s->newcol || !s->newins) { pwo_destroy(s); return NULL; }
    memset(s->w, 0, (size_t)s->cap * 6 * sizeof(uint64_t));
    for (int i = 0; i < W; i++) s->order[i] = i;
    for (int r = 0; r < T; r++) {
        const unsigned char *line = rows + (size_t)r * W;
        for (int i = 0; i < W; i++) {
            int c = code_of(line[i]);
            if (c < 0) { pwo_destroy(s); return NULL; }
            col_sym(s, i)[r] = (unsigned char)c;
            tally_add(col_w(s, i), c);
            if (c < 4) s->lengths[r]++;
        }
    }
    return s;
}

pwo_state *pwo_load(const char *path, int bandwidth, char *err, int errcap)
{
    FILE *f = fopen(path, "r");
    if (!f) { if (err) snprintf(err, errcap, "MA is missing."); return NULL; }      /* PW:121 */
    char *buf = malloc(MAX_LINE);
    unsigned char *rows = NULL;
    size_t cap = 0, used = 0;
    int T = 0, W = -1;
    if (!buf) { fclose(f); return NULL; }
    while (fgets(buf, MAX_LINE - 2, f)) {                                          /* PW:119-122 */
        size_t n = strlen(buf);
        if (n == 0 || buf[n - 1] != '\n') {                                         /* PW:134 */
            if (err) snprintf(err, errcap, "line %d is not terminated by a newline", T + 1);
            free(buf); free(rows); fclose(f); return NULL;
        }
        n--;
        if (W < 0) W = (int)n;
        if ((int)n != W) {   /* the reference silently corrupts its state here; we refuse */
            if (err) snprintf(err, errcap, "line %d has length %zu, expected %d", T + 1, n, W);
            free(buf); free(rows); fclose(f); return NULL;
        }
        if (used + n > cap) {
            cap = cap ? cap * 2 : (size_t)1 << 20;
            while (cap < used + n) cap *= 2;
            unsigned char *nr = realloc(rows, cap);
            if (!nr) { free(buf); free(rows); fclose(f); return NULL; }
            rows = nr;
        }
        memcpy(rows + used, buf, n);
        used += n;
        T++;
    }
    fclose(f);
    free(buf);
    pwo_state *s = (T > 0 && W > 0) ? pwo_create(T, W, rows, bandwidth) : NULL;
    if (!s && err) snprintf(err, errcap, "empty input, bad character or bandwidth out of range");
    free(rows);
    return s;
}

void pwo_destroy(pwo_state *s)
{
    if (!s) return;
    free(s->sym); free(s->w); free(s->order); free(s->scratch_order); free(s->freel);
    free(s->lengths); free(s->M); free(s->G); free(s->way); free(s->seqb); free(s->dbg_tallies);
    free(s->newcol); free(s->newins); free(s->pend_after); free(s->pend_slot);
    free(s);
}

int pwo_rows(const pwo_state *s) { return s->T; }
int pwo_width(const pwo_state *s) { return s->W; }
int pwo_row_length(const pwo_state *s, int k) { return s->lengths[k]; }
uint64_t pwo_cells(const pwo_state *s) { return s->cells; }

/* PW:459-645.  Forward pass: a '-' in the first surviving column, or whose row is blank in the
 * previous surviving column (W_Con2Columns[i-1], PW:495), becomes blank (tallies 0,1,2,3,5 -= 1);
 * a column left without a base is dropped on the spot (PW:515-549).  Backward pass: mirror image
 * with W_Con2Columns[i+1] (PW:556-634); every column it visits still holds a base, so it never
 * drops one. */
void pwo_trim(pwo_state *s)
{
    int n = 0;
    int *keep = s->scratch_order;
    for (int i = 0; i < s->W; i++) {
        int c = s->order[i];
        unsigned char *sy = col_sym(s, c);
        uint64_t *w = col_w(s, c);
        const unsigned char *pv = n > 0 ? col_sym(s, keep[n - 1]) : NULL;
        int has_base = 0;
        for (int k = 0; k < s->T; k++) {
            if (sy[k] == 4 && (pv == NULL || pv[k] == 5)) {
                sy[k] = 5;
                w[0]--; w[1]--; w[2]--; w[3]--; w[5]--;
            }
            if (sy[k] < 4) has_base = 1;
        }
        if (has_base) keep[n++] = c; else s->freel[s->nfree++] = c;
    }
    for (int i = n - 1; i >= 0; i--) {
        int c = keep[i];
        unsigned char *sy = col_sym(s, c);
        uint64_t *w = col_w(s, c);
        const unsigned char *nx = i < n - 1 ? col_sym(s, keep[i + 1]) : NULL;
        for (int k = 0; k < s->T; k++) {
            if (sy[k] == 4 && (nx == NULL || nx[k] == 5)) {
                sy[k] = 5;
                w[0]--; w[1]--; w[2]--; w[3]--; w[5]--;
            }
        }
    }
    memcpy(s->order, keep, (size_t)n * sizeof(int));
    s->W = n;
}

/* PW:706-763: unlink every column whose w[4] (number of real bases) is zero */
void pwo_compact(pwo_state *s)
{
    int n = 0;
    for (int i = 0; i < s->W; i++) {
        int c = s->order[i];
        if (col_w(s, c)[4] == 0) s->freel[s->nfree++] = c; else s->order[n++] = c;
    }
    s->W = n;
}

/* PW:765-859 (we check every column, the reference stops one short of the last) */
int pwo_check_tallies(const pwo_state *s)
{
    int bad = 0;
    for (int i = 0; i < s->W; i++) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        const unsigned char *sy = col_sym(s, s->order[i]);
        for (int k = 0; k < s->T; k++) tally_add(t, sy[k]);
        if (memcmp(t, col_w(s, s->order[i]), sizeof t) != 0) bad++;
    }
    return bad;
}

/* ---- the DP (PW:243-323) ---- */
static inline uint64_t S(const pwo_state *s, int y, int b) { return col_w(s, s->order[y])[b]; }   /* PW:243-246 */
static inline int anf_of(const pwo_state *s, int x) { int a = s->way[x] - s->H; return a > 0 ? a : 0; } /* PW:274 */
static inline uint64_t *Mrow(const pwo_state *s, int x) { return s->M + (size_t)x * s->B; }

/* PW:249-303 MatrixOut.  G[] holds prefix sums of S(.,4) taken before the traceback started;
 * the traceback only ever sums columns <= y, whose tallies are still untouched (PW:1222-1243 run
 * when the trace leaves a column), so the prefix sums equal the reference's running loop PW:289. */
static uint64_t Out(const pwo_state *s, int x, int y)
{
    if (x == -1) return 0;                               /* PW:256-263 */
    if (y == -1) return PWO_INF;                         /* PW:265-272 */
    int anf = anf_of(s, x);
    if (y - anf < 0) return PWO_INF;                     /* PW:276-283 */
    if (y - anf > s->B - 1) {                            /* PW:285-295 */
        uint64_t last = Mrow(s, x)[s->B - 1];
        if (x == s->L - 1) return last;
        return last + (s->G[y + 1] - s->G[anf + s->B]);
    }
    return Mrow(s, x)[y - anf];                          /* PW:302 */
}

static inline uint64_t umax(uint64_t a, uint64_t b) { return a > b ? a : b; }
static inline uint64_t umin(uint64_t a, uint64_t b) { return a < b ? a : b; }

/* PW:1222-1243 */
static void column_update(pwo_state *s, int y, int newsym, int k)
{
    int c = s->order[y];
    tally_add(col_w(s, c), newsym);
    col_sym(s, c)[k] = (unsigned char)newsym;
}

/* PW:1245-1332: open a new column directly after ordinal y (before any column opened there
 * earlier in this trace).  Other rows get '-' iff they are non-blank in column y and in the
 * column that followed y when the trace started, else blank: the reference looks at
 * New->Next, which may be a column opened earlier in this trace, but that one was filled by the
 * same rule from the same two original columns (PW:1305). */
static int column_add(pwo_state *s, int y, int base, int k)
{
    int ns = new_slot(s);
    if (ns < 0) return -1;
    unsigned char *nsy = col_sym(s, ns);
    uint64_t *nw = col_w(s, ns);
    uint64_t algap = 0;
    if (y == s->Wfill - 1) {                              /* PW:1286-1297 PreviousColumn==Last_Column */
        memset(nsy, 5, (size_t)s->T);
    } else {
        const unsigned char *a = col_sym(s, s->order[y]);
        const unsigned char *b = col_sym(s, s->order[y + 1]);
        for (int i = 0; i < s->T; i++) {
            if (i == k) continue;
            if (a[i] == 5 || b[i] == 5) nsy[i] = 5; else { nsy[i] = 4; algap++; }
        }
    }
    nsy[k] = (unsigned char)base;
    for (int i = 0; i < 6; i++) nw[i] = (i != base ? 1 : 0) + (i != 4 ? algap : 0);   /* PW:1320-1325 */
    s->pend_after[s->npend] = y;
    s->pend_slot[s->npend] = ns;
    s->npend++;
    return 0;
}

/* PW:1469-1531 Matrix_Filler + PW:1334-1454 Backtracker */
/* dev aid (scripts/dev/start_vectors.py): the fill of row k against the state AS IT IS -- TheWay, the row taken out of the
 * tallies, PW:1493-1513 -- without traceback or commit: the row is put back afterwards.  pwo_dbg_* then describe this fill. */
int pwo_fill_only(pwo_state *s, int k)
{
    pwo_compact(s);
    const int W = s->W, B = s->B;
    s->Wfill = W;
    int L = 0;
    for (int i = 0; i < W; i++) {
        int b = col_sym(s, s->order[i])[k];
        if (b < 4) { if (L > MAX_SEQ_LEN) return -2; s->way[L] = i; s->seqb[L] = (unsigned char)b; L++; }
    }
    s->L = L;
    if (L == 0) return 0;
    for (int i = 0; i < W; i++) tally_sub(col_w(s, s->order[i]), col_sym(s, s->order[i])[k]);
    for (int i = 0; i < W; i++) memcpy(s->dbg_tallies + (size_t)i * 6, col_w(s, s->order[i]), 6 * sizeof(uint64_t));
    s->G[0] = 0;
    for (int i = 0; i < W; i++) s->G[i + 1] = s->G[i] + S(s, i, 4);
    if ((size_t)L * B > s->Mcap) {
        free(s->M);
        s->Mcap = (size_t)L * B;
        s->M = malloc(s->Mcap * sizeof(uint64_t));
        if (!s->M) { s->Mcap = 0; return -1; }
    }
    /* (the reads of Out() only see the row's own symbols through the tallies, so the symbols may stay) */
    for (int x = 0; x < L; x++) {
        int anf = anf_of(s, x);
        int end = anf + B < W ? anf + B : W;
        uint64_t *row = Mrow(s, x);
        for (int y = anf; y < end; y++) {
            uint64_t e = Out(s, x - 1, y - 1) + S(s, y, s->seqb[x]);
            uint64_t left = (y - 1 < anf ? PWO_INF : row[y - 1 - anf]) + S(s, y, 4);
            e = umin(e, left);
            if (y > 0 && y < W - 1) e = umin(e, Out(s, x - 1, y) + umax(S(s, y, 5), S(s, y - 1, 5)));
            row[y - anf] = e;
        }
    }
    for (int i = 0; i < W; i++) tally_add(col_w(s, s->order[i]), col_sym(s, s->order[i])[k]);
    return 0;
}
const uint64_t *pwo_dbg_Mrow(const pwo_state *s, int x) { return Mrow(s, x); }

int pwo_realign_row(pwo_state *s, int k)
{
    pwo_compact(s);                                       /* PW:1478 */
    const int W = s->W, B = s->B, H = s->H;
    s->Wfill = W;
    /* PW:647-705 TheWay */
    int L = 0;
    for (int i = 0; i < W; i++) {
        int b = col_sym(s, s->order[i])[k];
        if (b < 4) {
            if (L > MAX_SEQ_LEN) return -2;               /* PW:675-680 */
            s->way[L] = i; s->seqb[L] = (unsigned char)b; L++;
        }
    }
    s->L = L;
    /* PW:1172-1220 remove the row from the profile */
    for (int i = 0; i < W; i++) {
        int c = s->order[i];
        unsigned char *sy = col_sym(s, c);
        tally_sub(col_w(s, c), sy[k]);
        sy[k] = 5;
    }
    if (L == 0) return 0;                                 /* PW:1488 */
    for (int i = 0; i < W; i++) memcpy(s->dbg_tallies + (size_t)i * 6, col_w(s, s->order[i]), 6 * sizeof(uint64_t));
    s->G[0] = 0;
    for (int i = 0; i < W; i++) s->G[i + 1] = s->G[i] + S(s, i, 4);
    if ((size_t)L * B > s->Mcap) {
        free(s->M);
        s->Mcap = (size_t)L * B;
        s->M = malloc(s->Mcap * sizeof(uint64_t));
        if (!s->M) { s->Mcap = 0; return -1; }
    }
    /* PW:1493-1513 fill */
    for (int x = 0; x < L; x++) {
        int anf = anf_of(s, x);
        int end = anf + B < W ? anf + B : W;              /* PW:1497 */
        uint64_t *row = Mrow(s, x);
        for (int y = anf; y < end; y++) {
            uint64_t e = Out(s, x - 1, y - 1) + S(s, y, s->seqb[x]);               /* PW:1503 */
            uint64_t left = (y - 1 < anf ? PWO_INF : row[y - 1 - anf]) + S(s, y, 4); /* PW:1504 */
            e = umin(e, left);
            if (y > 0 && y < W - 1)                                                /* PW:1505 */
                e = umin(e, Out(s, x - 1, y) + umax(S(s, y, 5), S(s, y - 1, 5)));  /* PW:1507 */
            row[y - anf] = e;                                                      /* PW:1510 */
        }
        s->cells += (uint64_t)(end - anf);
    }
    /* PW:1352-1360 entry column: strict '<' while scanning down keeps the largest y on ties */
    s->npend = 0;
    int x = L - 1, y = W - 1;
    int wayin = y;
    uint64_t best = Out(s, x, W - 1);
    {
        int lim = s->way[x] - H; if (lim < -1) lim = -1;
        while (y > lim) {
            uint64_t v = Out(s, x, y);
            if (v < best) { best = v; wayin = y; }
            y--;
        }
    }
    y = wayin;
    s->entry = wayin;
    /* PW:1362-1368: columns right of the entry: blank (row is already blank there) */
    /* PW:1371-1435 */
    while (x > -1 && y > -1) {
        uint64_t cur = Out(s, x, y);
        if (cur == Out(s, x, y - 1) + S(s, y, 4)) {                                /* PW:1375 (a) */
            column_update(s, y, x == L - 1 ? 5 : 4, k);
            y--;
        } else if (x == L - 1 && cur == Out(s, x, y - 1)) {                        /* PW:1386 (b) */
            column_update(s, y, 5, k);
            y--;
        } else if (cur == Out(s, x - 1, y - 1) + S(s, y, s->seqb[x])) {            /* PW:1394 (c) */
            column_update(s, y, s->seqb[x], k);
            s->newcol[x] = y; s->newins[x] = 0;
            x--; y--;
        } else if (y > 0 && cur == Out(s, x - 1, y) + umax(S(s, y, 5), S(s, y - 1, 5))) { /* PW:1404 (d) */
            if (column_add(s, y, s->seqb[x], k)) return -1;
            s->newcol[x] = y; s->newins[x] = 1;
            x--;
        } else {
            return -3;                                                             /* PW:1412-1427 */
        }
        if (Out(s, x, y) > 9046744073709551615ull) return -4;                      /* PW:1434 exit(0) */
    }
    /* PW:1437-1443 rest: blank (already) */
    if (x > -1) return -5;   /* cannot happen: leaving the loop with y == -1 and x >= 0 hits PW:1434 */
    /* splice the opened columns into the order: after ordinal y come its new columns in ascending x,
     * i.e. in reverse order of creation (each Column_Adder links directly behind column y, PW:1270-1274) */
    if (s->npend > 0) {
        int *no = s->scratch_order;
        int n = 0, p = s->npend - 1;     /* pend list is sorted by descending y (trace runs right to left) */
        for (int i = 0; i < W; i++) {
            no[n++] = s->order[i];
            while (p >= 0 && s->pend_after[p] == i) { no[n++] = s->pend_slot[p]; p--; }
        }
        memcpy(s->order, no, (size_t)n * sizeof(int));
        s->W = n;                                          /* PW:1268 Breite++ per new column */
    }
    return 0;
}

void pwo_realign_round(pwo_state *s)
{
    for (int k = 0; k < s->T; k++) pwo_realign_row(s, k);  /* PW:1695-1737 (k%500 W_Con is idempotent) */
}

/* PW:864-892 + PW:933-963: sum over rows of sum over non-blank cells of S(col, sym), i.e. per
 * column sum_b n_b * (cov - n_b) with n_b = cov - w[b] */
uint64_t pwo_total_score(pwo_state *s)
{
    pwo_compact(s);
    uint64_t total = 0;
    for (int i = 0; i < s->W; i++) {
        const uint64_t *w = col_w(s, s->order[i]);
        for (int b = 0; b < 5; b++) total += (w[5] - w[b]) * w[b];
    }
    return total;
}

void pwo_export(const pwo_state *s, unsigned char *out)
{
    static const char chars[6] = {'A', 'C', 'G', 'T', '-', ' '};                  /* PW:1558-1563 */
    for (int i = 0; i < s->W; i++) {
        const unsigned char *sy = col_sym(s, s->order[i]);
        for (int r = 0; r < s->T; r++) out[(size_t)r * s->W + i] = (unsigned char)chars[sy[r]];
    }
}

static int write_out(const pwo_state *s, const char *path)
{
    FILE *f = fopen(path, "w");                                                    /* PW:1566 */
    if (!f) return -1;
    unsigned char *buf = malloc((size_t)s->T * s->W + 1);
    if (!buf) { fclose(f); return -1; }
    pwo_export(s, buf);
    for (int r = 0; r < s->T; r++) {
        fwrite(buf + (size_t)r * s->W, 1, (size_t)s->W, f);
        fputc('\n', f);
    }
    free(buf);
    fclose(f);
    return 0;
}

/* PW:945-961: (millions, units) with units in (0, 1e6] */
static void print_score(FILE *log, uint64_t total)
{
    uint64_t m = 0, u = 0;
    if (total > 0) { m = (total - 1) / 1000000u; u = (total - 1) % 1000000u + 1; }
    fprintf(log, "OverallScore: %lu%06lu\n", (unsigned long)m, (unsigned long)u);
}

int pwo_run(const char *in_path, const char *out_path, int bandwidth, FILE *log, int max_rounds)
{
    char err[256];
    fprintf(log, "output file: %s\n", out_path);                                   /* PW:1649 */
    fprintf(log, "bandwidth %d\n", bandwidth);                                     /* PW:1650 */
    pwo_state *s = pwo_load(in_path, bandwidth, err, sizeof err);
    if (!s) { fprintf(log, "%s\n", err); return 1; }
    pwo_trim(s);                                                                   /* PW:1655 */
    fprintf(log, "Rows %d, Columns %d.\n", s->T, s->W);                            /* PW:1657 */
    pwo_compact(s);
    if (pwo_check_tallies(s)) { fprintf(log, "w_con_gau\n"); pwo_destroy(s); return 1; }
    uint64_t best = pwo_total_score(s);                                            /* PW:1664-1665 */
    print_score(log, best);
    clock_t t0 = clock();
    int rounds = 0;
    for (;;) {                                                                     /* PW:1681-1747 */
        if (max_rounds >= 0 && rounds >= max_rounds) break;
        if (rounds >= 10000) break;
        for (int k = 0; k < s->T; k++) {
            int rc = pwo_realign_row(s, k);
            if (rc == -4) { pwo_destroy(s); return 0; }
            if (rc < 0) { fprintf(log, "\nStuff gone wrong\n"); pwo_destroy(s); return 1; }
        }
        rounds++;
        uint64_t tot = pwo_total_score(s);
        print_score(log, tot);
        if (tot < best) {
            best = tot;
            if (write_out(s, out_path)) { fprintf(log, "DateiVerbratei!\n"); pwo_destroy(s); return 1; }
        } else break;
    }
    fprintf(log, "Total time: %f min.\n", ((double)(clock() - t0) / CLOCKS_PER_SEC) / 60);
    pwo_trim(s);                                                                   /* PW:1753 */
    uint64_t tot = pwo_total_score(s);
    print_score(log, tot);
    if (tot < best) {
        if (write_out(s, out_path)) { fprintf(log, "DateiVerbratei!\n"); pwo_destroy(s); return 1; }
    }
    pwo_destroy(s);
    return 0;
}

int pwo_dbg_L(const pwo_state *s) { return s->L; }
int pwo_dbg_W_at_fill(const pwo_state *s) { return s->Wfill; }
const int *pwo_dbg_way(const pwo_state *s) { return s->way; }
const unsigned char *pwo_dbg_seq(const pwo_state *s) { return s->seqb; }
const uint64_t *pwo_dbg_tallies(const pwo_state *s) { return s->dbg_tallies; }
const int *pwo_dbg_newcol(const pwo_state *s) { return s->newcol; }
const unsigned char *pwo_dbg_newins(const pwo_state *s) { return s->newins; }
int pwo_dbg_entry(const pwo_state *s) { return s->entry; }
uint64_t pwo_dbg_M(const pwo_state *s, int x, int j) { return Mrow(s, x)[j]; }

int pwo_row_columns(const pwo_state *s, int k, int *out, int cap)
{
    int n = 0;
    if (k < 0 || k >= s->T) return -1;
    for (int i = 0; i < s->W; i++)
        if (col_sym(s, s->order[i])[k] < 4) { if (n < cap) out[n] = i; n++; }
    return n;
}
