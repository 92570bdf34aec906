/*
 * phyoracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU-only restatement of physher's tree-likelihood hot path (SURVEY.md section 8a rows
 * P1, M1, M2, K1-K9, R1 and the per-category branch gradient K8).  It exists only to CHECK the HIP
 * path: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * physher_amd/ may.  It is pinned against the compiled reference through tests/golden/ (see
 * tests/test_oracle_golden.py).
 *
 * Array layouts are the reference's: partials [C][P][S], matrices [C][S][S] row-major
 * P[parent state][child state], tip states uint8 [T][P] with codes >= S meaning "unknown".
 */
#ifndef PHYORACLE_H
#define PHYORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PHYO_NUCLEOTIDE = 0, PHYO_AMINO_ACID = 1, PHYO_CODON = 2 };

/* State code of one symbol (datatype.c:55-89,406-408; codon: sitepattern.c:808-819). `sym` points at
 * 1 character (3 for codons). */
int phyo_encode_symbol(int datatype, const char *sym);

/* Ambiguity mask of a nucleotide code (datatype.h:26-66); generic one-hot / all-ones otherwise
 * (datatype.c:212-240). */
void phyo_state_partial(int datatype, int state_count, int code, double *partial);

/* P1: column de-duplication in the reference's hashtable iteration order
 * (sitepattern.c:71-79,186-251,731-754; hashtable.c:31,119-151,188-249,262-308,414-451).
 * columns: [site_count][taxon_count] state codes.  Outputs: patterns [taxon_count][*pattern_count]
 * (caller allocates taxon_count*site_count), weights [site_count].  Returns 0. */
int phyo_compress_patterns(const uint8_t *columns, int site_count, int taxon_count, uint8_t *patterns,
                           double *weights, int *pattern_count);

/* M1 / M2: P(t) = |U diag(exp(lambda t)) U^-1| (substmodel.c:518-557), dP/dt (substmodel.c:695-723). */
void phyo_p_t(int S, const double *eval, const double *evec, const double *ivec, double t, double *P);
void phyo_dp_dt(int S, const double *eval, const double *evec, const double *ivec, double t, double *P);

typedef struct {
	int tip_count, node_count, pattern_count, state_count, cat_count;
	const int32_t *left, *right; /* [N] children, -1 for tips; ids as tree.c:183-224 */
	int root;
	const uint8_t *tip_states;  /* [T][P]; used when tip_partials == NULL ("tipstates": true semantics) */
	const double *tip_partials; /* [T][P][S] or NULL ("tipstates": false semantics, treelikelihood.c:1106-1117) */
	const double *weights;      /* [P] */
	const double *eval, *evec, *ivec; /* eigen system [S], [S][S], [S][S] */
	const double *freqs;        /* [S] */
	const double *cat_rates, *cat_props; /* [C] */
	const double *branch_lengths;        /* [N], root ignored */
	int rescale; /* 0: never; 1: always (SingleTreeLikelihood_use_rescaling); 2: lazily on +-inf (treelikelihood.c:1496-1519) */
	int compat_scaled_gradient; /* 1: per-category denominators under rescaling like treelikelihood.c:2851-2870 */
	int fold_root_freqs; /* 1: pi multiplied into the uppers of the root's children (include_root_freqs = true,
	                        treelikelihood.c:241,2147-2153: what the reference does when only tree/site/clock gradients are
	                        requested -- only valid for uniform pi, see tests/test_oracle_golden.py);
	                        0: pi applied in the final state sum (include_root_freqs = false, :294-305, :2715-2751) */
} phyo_problem;

/* Lower pass + root integration (treelikelihood.c:1454-1526). pattern_lk [P].
 * lower: optional [N][C][P][S] (tips filled too); scaling: optional [N][P] cumulative log factors.
 * Returns lnL; *rescaled tells whether rescaling ended up on. */
double phyo_log_likelihood(const phyo_problem *pb, double *pattern_lk, double *lower, double *scaling, int *rescaled);

/* Upper pass + per-category branch gradient (treelikelihood.c:2129-2161, 2793-2941).
 * cat_grad [N][C] (root row untouched = 0). upper: optional [N][C][P][S].  Returns lnL. */
double phyo_gradient(const phyo_problem *pb, double *pattern_lk, double *cat_grad, double *lower, double *upper, int *rescaled);

#ifdef __cplusplus
}
#endif
#endif
