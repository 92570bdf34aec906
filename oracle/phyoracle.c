/*
 * phyoracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see phyoracle.h).
 *
 * CPU restatement of physher's Felsenstein-pruning likelihood and per-category branch gradient.
 * Written from the algorithm, flat arrays and an explicit post-order schedule instead of the
 * reference's recursion and function-pointer kernels; each function cites the reference lines it
 * follows.  Pinned against the compiled reference by tests/golden/.
 */
#include "phyoracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* encodings                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* datatype.c:74-89 (NUCLEOTIDE_STATES): A C G T/U =0..3, R Y M W S K = 5..10, B D H V N = 11..15,
 * '?' and unknown letters = 16, '-' and everything else = 17. */
static int nuc_code(unsigned char ch) {
	switch (ch) {
		case 'A': case 'a': return 0;
		case 'C': case 'c': return 1;
		case 'G': case 'g': return 2;
		case 'T': case 't': case 'U': case 'u': return 3;
		case 'R': case 'r': return 5;
		case 'Y': case 'y': return 6;
		case 'M': case 'm': return 7;
		case 'W': case 'w': return 8;
		case 'S': case 's': return 9;
		case 'K': case 'k': return 10;
		case 'B': case 'b': return 11;
		case 'D': case 'd': return 12;
		case 'H': case 'h': return 13;
		case 'V': case 'v': return 14;
		case 'N': case 'n': return 15;
		case '?': return 16;
		default:
			if ((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z')) return 16;
			return 17;
	}
}

/* datatype.c:55-70 (AMINO_ACID_STATES), alphabet "ACDEFGHIKLMNPQRSTVWYBZX*?-". */
static int aa_code(unsigned char ch) {
	static const char *alpha = "ACDEFGHIKLMNPQRSTVWY";
	if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 'a' + 'A');
	for (int i = 0; i < 20; i++)
		if (alpha[i] == (char)ch) return i;
	switch (ch) {
		case 'B': return 20;
		case 'Z': return 21;
		case 'X': return 22;
		case '*': return 23;
		case 'J': case 'O': case 'U': case '?': return 24;
		default: return 25;
	}
}

int phyo_encode_symbol(int datatype, const char *sym) {
	if (datatype == PHYO_NUCLEOTIDE) return nuc_code((unsigned char)sym[0]);
	if (datatype == PHYO_AMINO_ACID) return aa_code((unsigned char)sym[0]);
	/* codons (sitepattern.c:796-819): n1*16+n2*4+n3 minus the number of stop codons before it in the
	 * universal code (TAA=48, TAG=50, TGA=56); any non-ACGT position -> 65. */
	int n1 = nuc_code((unsigned char)sym[0]), n2 = nuc_code((unsigned char)sym[1]), n3 = nuc_code((unsigned char)sym[2]);
	if (n1 > 3 || n2 > 3 || n3 > 3) return 65;
	int v = n1 * 16 + n2 * 4 + n3, code = v;
	if (v > 48) code--;
	if (v > 50) code--;
	if (v > 56) code--;
	return code;
}

void phyo_state_partial(int datatype, int S, int code, double *partial) {
	if (datatype == PHYO_NUCLEOTIDE) {
		/* datatype.h:26-66 */
		static const unsigned char mask[18] = {1, 2, 4, 8, 8, 5, 10, 3, 9, 6, 12, 14, 13, 11, 7, 15, 15, 15};
		unsigned char m = mask[code < 18 ? code : 17];
		for (int i = 0; i < 4; i++) partial[i] = (m >> i) & 1 ? 1.0 : 0.0;
		return;
	}
	for (int i = 0; i < S; i++) partial[i] = code >= S ? 1.0 : 0.0;
	if (code < S) partial[code] = 1.0;
}

/* ------------------------------------------------------------------------------------------ */
/* P1: pattern compression                                                                     */
/* ------------------------------------------------------------------------------------------ */

typedef struct entry {
	const uint8_t *key;
	unsigned int hash;
	int count;
	struct entry *next;
} entry_t;

static const unsigned int PRIMES[] = {5,        53,       97,        193,       389,       769,       1543,      3079,     6151,
                                      12289,    24593,    49157,     98317,     196613,    393241,    786433,    1572869,  3145739,
                                      6291469,  12582917, 25165843,  50331653,  100663319, 201326611, 402653189, 805306457, 1610612741};

/* sitepattern.c:71-79 then the Java-1.4 mixer of hashtable.c:188-197, all in 32-bit unsigned. */
static unsigned int column_hash(const uint8_t *v, int n) {
	unsigned int h = v[0];
	for (int i = 1; i < n; i++) h ^= v[i] + 0x9e3779b9u + (h << 6) + (h >> 2);
	h += ~(h << 9);
	h ^= ((h >> 14) | (h << 18));
	h += (h << 4);
	h ^= ((h >> 10) | (h << 22));
	return h;
}

int phyo_compress_patterns(const uint8_t *columns, int site_count, int T, uint8_t *patterns, double *weights, int *pattern_count) {
	int pidx = 3; /* new_Hashtable(100, ...): first prime >= 100 (hashtable.c:119-127) */
	unsigned int size = PRIMES[pidx];
	unsigned int limit = (unsigned int)ceil(size * 0.65);
	unsigned int length = 0;
	entry_t **table = calloc(size, sizeof(entry_t *));
	entry_t *pool = malloc(sizeof(entry_t) * (size_t)(site_count > 0 ? site_count : 1));
	for (int s = 0; s < site_count; s++) {
		const uint8_t *col = columns + (size_t)s * T;
		unsigned int h = column_hash(col, T);
		entry_t *e = table[h % size];
		for (; e; e = e->next)
			if (e->hash == h && memcmp(e->key, col, (size_t)T) == 0) break; /* hashtable.c:319-321 */
		if (e) {
			e->count++;
			continue;
		}
		if (length == limit) { /* Hashtable_add -> Hashtable_expand (hashtable.c:262-266,199-249) */
			unsigned int newsize = PRIMES[++pidx];
			entry_t **nt = calloc(newsize, sizeof(entry_t *));
			for (unsigned int i = 0; i < size; i++) {
				entry_t *x;
				while ((x = table[i]) != NULL) { /* pop head, push on head of new chain: chains reverse */
					table[i] = x->next;
					unsigned int idx = x->hash % newsize;
					x->next = nt[idx];
					nt[idx] = x;
				}
			}
			free(table);
			table = nt;
			size = newsize;
			limit = (unsigned int)ceil(size * 0.65);
		}
		e = &pool[length];
		e->key = col;
		e->hash = h;
		e->count = 1;
		e->next = table[h % size]; /* new keys are pushed on the head (hashtable.c:300-303) */
		table[h % size] = e;
		length++;
	}
	/* iteration: buckets 0..size-1, each chain head -> tail (hashtable.c:414-451) */
	int P = (int)length, k = 0;
	for (unsigned int i = 0; i < size; i++)
		for (entry_t *e = table[i]; e; e = e->next) {
			weights[k] = e->count;
			for (int t = 0; t < T; t++) patterns[(size_t)t * P + k] = e->key[t];
			k++;
		}
	*pattern_count = P;
	free(pool);
	free(table);
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* M1 / M2                                                                                     */
/* ------------------------------------------------------------------------------------------ */

void phyo_p_t(int S, const double *eval, const double *evec, const double *ivec, double t, double *P) {
	double *tmp = malloc(sizeof(double) * S * S);
	for (int i = 0; i < S; i++) {
		double e = exp(eval[i] * t);
		for (int j = 0; j < S; j++) tmp[i * S + j] = ivec[i * S + j] * e;
	}
	for (int i = 0; i < S; i++)
		for (int j = 0; j < S; j++) {
			double s = 0.;
			for (int k = 0; k < S; k++) s += tmp[k * S + j] * evec[i * S + k];
			P[i * S + j] = fabs(s); /* substmodel.c:552 */
		}
	free(tmp);
}

void phyo_dp_dt(int S, const double *eval, const double *evec, const double *ivec, double t, double *P) {
	double *tmp = malloc(sizeof(double) * S * S);
	for (int i = 0; i < S; i++) {
		double e = eval[i] * exp(eval[i] * t);
		for (int j = 0; j < S; j++) tmp[i * S + j] = ivec[i * S + j] * e;
	}
	for (int i = 0; i < S; i++)
		for (int j = 0; j < S; j++) {
			double s = 0.;
			for (int k = 0; k < S; k++) s += tmp[k * S + j] * evec[i * S + k];
			P[i * S + j] = s; /* no fabs: substmodel.c:712-720 */
		}
	free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* likelihood                                                                                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
	const phyo_problem *pb;
	int *postorder; /* internal nodes, children before parents */
	int *parent;
	double *lower;   /* [N][C][P][S] */
	double *upper;   /* [N][C][P][S] or NULL */
	double *mats;    /* [N][C][S][S] */
	double *scale;   /* [2N][P] cumulative log factors (lower: n, upper: N+n) or NULL */
	size_t psz;      /* C*P*S */
} work_t;

static void build_schedule(work_t *w) {
	const phyo_problem *pb = w->pb;
	int N = pb->node_count;
	w->postorder = malloc(sizeof(int) * N);
	w->parent = malloc(sizeof(int) * N);
	for (int i = 0; i < N; i++) w->parent[i] = -1;
	int *stack = malloc(sizeof(int) * 2 * N), *state = calloc(N, sizeof(int));
	int sp = 0, n_out = 0;
	stack[sp++] = pb->root;
	while (sp) {
		int n = stack[sp - 1];
		if (pb->left[n] < 0) { sp--; continue; }
		if (state[n] == 0) {
			state[n] = 1;
			w->parent[pb->left[n]] = n;
			w->parent[pb->right[n]] = n;
			stack[sp++] = pb->right[n];
			stack[sp++] = pb->left[n];
		} else {
			sp--;
			w->postorder[n_out++] = n;
		}
	}
	w->postorder[n_out] = -1;
	free(stack);
	free(state);
}

static void fill_tips(work_t *w) {
	const phyo_problem *pb = w->pb;
	int P = pb->pattern_count, S = pb->state_count, C = pb->cat_count;
	for (int t = 0; t < pb->tip_count; t++) {
		double *dst = w->lower + (size_t)t * w->psz;
		for (int k = 0; k < P; k++) {
			if (pb->tip_partials) memcpy(dst + (size_t)k * S, pb->tip_partials + ((size_t)t * P + k) * S, sizeof(double) * S);
			else phyo_state_partial(-1, S, pb->tip_states[(size_t)t * P + k], dst + (size_t)k * S);
		}
		for (int c = 1; c < C; c++) memcpy(dst + (size_t)c * P * S, dst, sizeof(double) * P * S); /* treelikelihood.c:1111-1114 */
	}
}

static void fill_matrices(work_t *w) {
	const phyo_problem *pb = w->pb;
	int S = pb->state_count, C = pb->cat_count;
	for (int n = 0; n < pb->node_count; n++) {
		if (n == pb->root) continue;
		for (int c = 0; c < C; c++) /* treelikelihood.c:1672-1696: t = bl * rate_c */
			phyo_p_t(S, pb->eval, pb->evec, pb->ivec, pb->branch_lengths[n] * pb->cat_rates[c], w->mats + ((size_t)n * C + c) * S * S);
	}
}

/* out[c][k][i] = (sum_j M1[c][i][j] p1[c][k][j]) * (sum_j M2[c][i][j] p2[c][k][j])   (K2, treelikelihood4.c:1160-1277,
 * generic treelikelihoodX.c:166-576).  p2 == NULL: single-child form (K4, partials_undefined). */
static void combine(const phyo_problem *pb, const double *p1, const double *M1, const double *p2, const double *M2, double *out) {
	int P = pb->pattern_count, S = pb->state_count, C = pb->cat_count;
	for (int c = 0; c < C; c++)
		for (int k = 0; k < P; k++) {
			const double *a = p1 + ((size_t)c * P + k) * S;
			const double *b = p2 ? p2 + ((size_t)c * P + k) * S : NULL;
			double *o = out + ((size_t)c * P + k) * S;
			for (int i = 0; i < S; i++) {
				double s1 = 0., s2 = 0.;
				const double *m1 = M1 + ((size_t)c * S + i) * S;
				for (int j = 0; j < S; j++) s1 += m1[j] * a[j];
				if (b) {
					const double *m2 = M2 + ((size_t)c * S + i) * S;
					for (int j = 0; j < S; j++) s2 += m2[j] * b[j];
					o[i] = s1 * s2;
				} else
					o[i] = s1;
			}
		}
}

/* R1: SingleTreeLikelihood_scalePartials (treelikelihood.c:1790-1836). sf1/sf2 may be NULL. */
static void scale_partials(const phyo_problem *pb, double *partials, double *sf, const double *sf1, const double *sf2) {
	int P = pb->pattern_count, S = pb->state_count, C = pb->cat_count;
	for (int k = 0; k < P; k++) {
		double m = 0.0;
		for (int c = 0; c < C; c++)
			for (int j = 0; j < S; j++) {
				double v = partials[((size_t)c * P + k) * S + j];
				if (v > m) m = v;
			}
		if (m < 1.E-40) {
			for (int c = 0; c < C; c++)
				for (int j = 0; j < S; j++) partials[((size_t)c * P + k) * S + j] /= m;
			sf[k] = log(m);
		} else
			sf[k] = 0.0;
		if (sf1) sf[k] += sf1[k];
		if (sf2) sf[k] += sf2[k];
	}
}

static double lower_pass(work_t *w, double *pattern_lk, int scale_on) {
	const phyo_problem *pb = w->pb;
	int P = pb->pattern_count, S = pb->state_count, C = pb->cat_count, T = pb->tip_count;
	for (int q = 0; w->postorder[q] >= 0; q++) {
		int n = w->postorder[q], l = pb->left[n], r = pb->right[n];
		combine(pb, w->lower + (size_t)l * w->psz, w->mats + (size_t)l * C * S * S, w->lower + (size_t)r * w->psz, w->mats + (size_t)r * C * S * S,
		        w->lower + (size_t)n * w->psz);
		if (scale_on)
			scale_partials(pb, w->lower + (size_t)n * w->psz, w->scale + (size_t)n * P, l >= T ? w->scale + (size_t)l * P : NULL,
			               r >= T ? w->scale + (size_t)r * P : NULL);
	}
	/* K5 integrate (treelikelihood4.c:822-880), K6 log (treelikelihood4.c:882-915), sum (treelikelihood.c:1482-1487) */
	const double *rootp = w->lower + (size_t)pb->root * w->psz;
	double lnl = 0;
	for (int k = 0; k < P; k++) {
		double L = 0;
		for (int s = 0; s < S; s++) {
			double r = 0;
			for (int c = 0; c < C; c++) r += pb->cat_props[c] * rootp[((size_t)c * P + k) * S + s];
			L += pb->freqs[s] * r;
		}
		pattern_lk[k] = log(L);
		if (scale_on) pattern_lk[k] += w->scale[(size_t)pb->root * P + k];
		lnl += pattern_lk[k] * pb->weights[k];
	}
	return lnl;
}

static void work_init(work_t *w, const phyo_problem *pb, int want_upper) {
	memset(w, 0, sizeof(*w));
	w->pb = pb;
	w->psz = (size_t)pb->cat_count * pb->pattern_count * pb->state_count;
	w->lower = calloc((size_t)pb->node_count * w->psz, sizeof(double));
	if (want_upper) w->upper = calloc((size_t)pb->node_count * w->psz, sizeof(double));
	w->mats = calloc((size_t)pb->node_count * pb->cat_count * pb->state_count * pb->state_count, sizeof(double));
	w->scale = calloc((size_t)2 * pb->node_count * pb->pattern_count, sizeof(double));
	build_schedule(w);
	fill_tips(w);
	fill_matrices(w);
}

static void work_free(work_t *w) {
	free(w->lower);
	free(w->upper);
	free(w->mats);
	free(w->scale);
	free(w->postorder);
	free(w->parent);
}

static double run_lower(work_t *w, double *pattern_lk, int *scale_on) {
	const phyo_problem *pb = w->pb;
	*scale_on = pb->rescale == 1;
	double lnl = lower_pass(w, pattern_lk, *scale_on);
	if (isinf(lnl) && pb->rescale == 2) { /* treelikelihood.c:1496-1519 */
		*scale_on = 1;
		lnl = lower_pass(w, pattern_lk, 1);
	}
	return lnl;
}

double phyo_log_likelihood(const phyo_problem *pb, double *pattern_lk, double *lower, double *scaling, int *rescaled) {
	work_t w;
	work_init(&w, pb, 0);
	int on;
	double lnl = run_lower(&w, pattern_lk, &on);
	if (lower) memcpy(lower, w.lower, sizeof(double) * (size_t)pb->node_count * w.psz);
	if (scaling) memcpy(scaling, w.scale, sizeof(double) * (size_t)pb->node_count * pb->pattern_count);
	if (rescaled) *rescaled = on;
	work_free(&w);
	return lnl;
}

double phyo_gradient(const phyo_problem *pb, double *pattern_lk, double *cat_grad, double *lower, double *upper, int *rescaled) {
	int P = pb->pattern_count, S = pb->state_count, C = pb->cat_count, N = pb->node_count, T = pb->tip_count;
	work_t w;
	work_init(&w, pb, 1);
	int on;
	double lnl = run_lower(&w, pattern_lk, &on);
	if (rescaled) *rescaled = on;
	memset(cat_grad, 0, sizeof(double) * (size_t)N * C);
	if (isnan(lnl) || isinf(lnl)) { /* treelikelihood.c:327-332 */
		for (size_t i = 0; i < (size_t)N * C; i++) cat_grad[i] = NAN;
		work_free(&w);
		return lnl;
	}
	size_t msz = (size_t)C * S * S;
	/* K7 upper pass, pre-order = reverse of the post-order list (treelikelihood.c:2129-2161) */
	int n_int = 0;
	while (w.postorder[n_int] >= 0) n_int++;
	for (int q = n_int - 1; q >= 0; q--) {
		int p = w.postorder[q];
		for (int side = 0; side < 2; side++) {
			int n = side ? pb->right[p] : pb->left[p];
			int s = side ? pb->left[p] : pb->right[p];
			double *un = w.upper + (size_t)n * w.psz;
			double *sfn = w.scale + (size_t)(N + n) * P;
			const double *sfs = s >= T ? w.scale + (size_t)s * P : NULL;
			if (p == pb->root) {
				combine(pb, w.lower + (size_t)s * w.psz, w.mats + (size_t)s * msz, NULL, NULL, un); /* u_n = P_s p_s */
				if (on) scale_partials(pb, un, sfn, sfs, NULL);
				if (pb->fold_root_freqs)
					for (size_t i = 0; i < w.psz; i++) un[i] *= pb->freqs[i % S]; /* :2148-2153, include_root_freqs */
			} else {
				combine(pb, w.upper + (size_t)p * w.psz, w.mats + (size_t)p * msz, w.lower + (size_t)s * w.psz, w.mats + (size_t)s * msz, un);
				if (on) scale_partials(pb, un, sfn, w.scale + (size_t)(N + p) * P, sfs);
			}
		}
	}
	/* K8 per-category branch gradient (treelikelihood.c:2793-2941, treelikelihood4.c:1633-1775) */
	double *dP = malloc(sizeof(double) * msz);
	double *num = malloc(sizeof(double) * (size_t)C * P), *den = malloc(sizeof(double) * (size_t)C * P);
	for (int n = 0; n < N; n++) {
		if (n == pb->root) continue;
		for (int c = 0; c < C; c++) phyo_dp_dt(S, pb->eval, pb->evec, pb->ivec, pb->branch_lengths[n] * pb->cat_rates[c], dP + (size_t)c * S * S);
		const double *un = w.upper + (size_t)n * w.psz, *pn = w.lower + (size_t)n * w.psz;
		for (int c = 0; c < C; c++)
			for (int k = 0; k < P; k++) {
				double a = 0, b = 0;
				for (int i = 0; i < S; i++) {
					double s1 = 0, s2 = 0;
					for (int j = 0; j < S; j++) {
						s1 += dP[((size_t)c * S + i) * S + j] * pn[((size_t)c * P + k) * S + j];
						s2 += w.mats[(size_t)n * msz + ((size_t)c * S + i) * S + j] * pn[((size_t)c * P + k) * S + j];
					}
					double f = pb->fold_root_freqs ? 1.0 : pb->freqs[i]; /* :2743-2747 vs :2781-2785 */
					a += f * un[((size_t)c * P + k) * S + i] * s1;
					b += f * un[((size_t)c * P + k) * S + i] * s2;
				}
				num[(size_t)c * P + k] = a;
				den[(size_t)c * P + k] = b;
			}
		for (int c = 0; c < C; c++) {
			double g = 0;
			for (int k = 0; k < P; k++) {
				double L;
				if (!on) L = exp(pattern_lk[k]); /* treelikelihood.c:3207-3210 */
				else if (pb->compat_scaled_gradient) L = den[(size_t)c * P + k]; /* :2851-2870: per-category denominator */
				else {
					L = 0; /* site likelihood in the scaled units of this branch */
					for (int cc = 0; cc < C; cc++) L += pb->cat_props[cc] * den[(size_t)cc * P + k];
				}
				g += num[(size_t)c * P + k] / L * pb->weights[k];
			}
			cat_grad[(size_t)n * C + c] = g;
		}
	}
	free(dP);
	free(num);
	free(den);
	if (lower) memcpy(lower, w.lower, sizeof(double) * (size_t)N * w.psz);
	if (upper) memcpy(upper, w.upper, sizeof(double) * (size_t)N * w.psz);
	work_free(&w);
	return lnl;
}
