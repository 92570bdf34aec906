/*
 * ref_driver.c -- TEST INFRASTRUCTURE (oracle side).
 *
 * A small program of this repository's own that links against the *compiled reference*
 * (oracle/_ref/libphyc_ref.so, built by oracle/Makefile from the sources where they lie under
 * /root/reference) and dumps golden vectors / timings for the tree-likelihood hot path.
 * It is only ever built where /root/reference exists; nothing from the reference is copied.
 *
 * It mirrors the call protocol of the reference's own harness and wrapper:
 *   - model assembly as examples/benchmarking.c:426-460 (patterns + tree in a hash, likelihood
 *     node from JSON) and src/phycpp/physher.cpp:34-49 (new_TreeModel_from_newick with the taxa
 *     in alignment order, so tip id == sequence index);
 *   - timing protocol as examples/benchmarking.c:466-471 and :498-503.
 *
 * Usage:
 *   ref_driver dump  <spec-file> <out.json>
 *   ref_driver json  <reference-json-file> <out.json>      (run in the dir holding its data files)
 *   ref_driver bench <spec-file> <iters> <warmup>
 *   ref_driver brent <spec-file> <out.json>     the optimiser's single-branch call pattern (optimizer.c:112-153)
 *   ref_driver attr  <attr-spec-file> <out.json>           (one discrete trait per taxon, any state count)
 *   ref_driver branch <spec-file> <out.json>               (lnL, d lnL/dt, d2 lnL/dt2 of single branches at trial lengths)
 *
 * spec-file: "key value" lines:
 *   fasta <path>        newick <path-to-file-with-newick>
 *   datatype nucleotide|aa|codon
 *   model jc69|hky|gtr|wag|lg|mg94
 *   rates a,b,c,...     (gtr: ac,ag,at,cg,ct ; hky: kappa ; mg94: kappa,alpha,beta)
 *   freqs f0,f1,...     (omit for the empirical aa models -> model frequencies)
 *   categories C        alpha A   (C>1 => discrete gamma)
 *   tipstates 0|1       sse 0|1   rescale 0|1
 *   generic_kernels 0|1 (force the generic-state kernels; needed for 61 states, SURVEY 8a note)
 *
 * attr-spec-file (the model src/phycpp/physher.cpp:594-629 assembles: GeneralDataType + AttributePattern +
 * GeneralSubstitutionModel): "key value" lines:
 *   states s0,s1,...          ambiguity NAME=s0|s3   (repeatable)
 *   traits <path>             lines "taxon value", tip id == line index
 *   newick <path>             structure i,j,...  (rate index of each upper-, then each lower-triangle entry)
 *   rates r0,r1,...           freqs f0,...       normalize 0|1     categories C   alpha A
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "phyc/datatype.h"
#include "phyc/discreteparameter.h"
#include "phyc/filereader.h"
#include "phyc/hashtable.h"
#include "phyc/mjson.h"
#include "phyc/mg94.h"
#include "phyc/parameters.h"
#include "phyc/sequenceio.h"
#include "phyc/simplex.h"
#include "phyc/sitemodel.h"
#include "phyc/sitepattern.h"
#include "phyc/substmodel.h"
#include "phyc/tree.h"
#include "phyc/treelikelihood.h"
#include "phyc/treelikelihoodX.h"

typedef struct {
	char fasta[1024], newick[1024], datatype[32], model[32];
	double rates[16];
	int nrates;
	double freqs[80];
	int nfreqs;
	int categories;
	double alpha;
	int tipstates, sse, rescale, generic_kernels;
	char sitedist[32]; /* gamma | weibull | discrete (+I only) */
	double pinv, mu;   /* < 0: absent */
} spec_t;

static int parse_list(const char *s, double *out, int max) {
	int n = 0;
	char *tmp = strdup(s), *tok, *save;
	for (tok = strtok_r(tmp, ",", &save); tok && n < max; tok = strtok_r(NULL, ",", &save)) out[n++] = atof(tok);
	free(tmp);
	return n;
}

static void read_spec(const char *path, spec_t *sp) {
	memset(sp, 0, sizeof(*sp));
	strcpy(sp->datatype, "nucleotide");
	strcpy(sp->model, "jc69");
	sp->categories = 1;
	sp->alpha = 0.5;
	sp->tipstates = 0;
	sp->sse = 1;
	strcpy(sp->sitedist, "gamma");
	sp->pinv = -1;
	sp->mu = -1;
	FILE *f = fopen(path, "r");
	if (!f) { fprintf(stderr, "cannot open spec %s\n", path); exit(2); }
	char key[64], val[4096];
	while (fscanf(f, "%63s %4095s", key, val) == 2) {
		if (!strcmp(key, "fasta")) strcpy(sp->fasta, val);
		else if (!strcmp(key, "newick")) strcpy(sp->newick, val);
		else if (!strcmp(key, "datatype")) strcpy(sp->datatype, val);
		else if (!strcmp(key, "model")) strcpy(sp->model, val);
		else if (!strcmp(key, "rates")) sp->nrates = parse_list(val, sp->rates, 16);
		else if (!strcmp(key, "freqs")) sp->nfreqs = parse_list(val, sp->freqs, 80);
		else if (!strcmp(key, "categories")) sp->categories = atoi(val);
		else if (!strcmp(key, "alpha")) sp->alpha = atof(val);
		else if (!strcmp(key, "tipstates")) sp->tipstates = atoi(val);
		else if (!strcmp(key, "sse")) sp->sse = atoi(val);
		else if (!strcmp(key, "rescale")) sp->rescale = atoi(val);
		else if (!strcmp(key, "generic_kernels")) sp->generic_kernels = atoi(val);
		else if (!strcmp(key, "sitedist")) strcpy(sp->sitedist, val);
		else if (!strcmp(key, "pinv")) sp->pinv = atof(val);
		else if (!strcmp(key, "mu")) sp->mu = atof(val);
		else { fprintf(stderr, "unknown spec key %s\n", key); exit(2); }
	}
	fclose(f);
}

static char *slurp(const char *path) {
	FILE *f = fopen(path, "r");
	if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *b = malloc(n + 1);
	if (fread(b, 1, n, f) != (size_t)n) { fprintf(stderr, "short read\n"); exit(2); }
	b[n] = 0;
	while (n > 0 && (b[n - 1] == '\n' || b[n - 1] == '\r' || b[n - 1] == ' ')) b[--n] = 0;
	fclose(f);
	return b;
}

typedef struct {
	Model *mlike;
	Model *mtree;
	SitePattern *patterns;
	Hashtable *hash;
	json_node *json;
} built_t;

static void append(char **buf, size_t *len, size_t *cap, const char *fmt, ...) {
	va_list ap;
	for (;;) {
		va_start(ap, fmt);
		int n = vsnprintf(*buf + *len, *cap - *len, fmt, ap);
		va_end(ap);
		if ((size_t)n < *cap - *len) { *len += n; return; }
		*cap = (*cap + n) * 2;
		*buf = realloc(*buf, *cap);
	}
}

static built_t build_from_spec(const spec_t *sp) {
	built_t b;
	memset(&b, 0, sizeof(b));
	b.hash = new_Hashtable_string(100);
	hashtable_set_key_ownership(b.hash, false);
	hashtable_set_value_ownership(b.hash, false);

	Sequences *sequences = readSequences(sp->fasta);
	bool codon = !strcmp(sp->datatype, "codon");
	if (codon) sequences->datatype = new_CodonDataType(0);
	else if (!strcmp(sp->datatype, "aa")) sequences->datatype = new_AminoAcidDataType();
	else sequences->datatype = new_NucleotideDataType();
	b.patterns = new_SitePattern(sequences);
	int ntaxa = sequences->size;
	char **taxa = malloc(sizeof(char *) * ntaxa);
	for (int i = 0; i < ntaxa; i++) taxa[i] = strdup(b.patterns->names[i]);
	free_Sequences(sequences);
	Hashtable_add(b.hash, "patterns", b.patterns);

	char *newick = slurp(sp->newick);
	b.mtree = new_TreeModel_from_newick(newick, taxa, NULL);
	free(newick);
	Hashtable_add(b.hash, "tree", b.mtree);

	size_t cap = 8192, len = 0;
	char *js = malloc(cap);
	js[0] = 0;
	append(&js, &len, &cap, "{\"id\":\"treelikelihood\",\"type\":\"treelikelihood\",\"sse\":%s,\"tipstates\":%s,\"sitepattern\":\"&patterns\",\"tree\":\"&tree\",",
	       sp->sse ? "true" : "false", sp->tipstates ? "true" : "false");
	/* site model (sitemodel.c:1056-1246): "categories" counts the variable classes; a 2-simplex of proportions adds +I */
	append(&js, &len, &cap, "\"sitemodel\":{\"id\":\"sitemodel\",\"type\":\"sitemodel\"");
	if (sp->categories > 1 || sp->pinv >= 0) {
		const bool discrete = !strcmp(sp->sitedist, "discrete");
		append(&js, &len, &cap, ",\"distribution\":{\"distribution\":\"%s\"", sp->sitedist);
		if (!discrete)
			append(&js, &len, &cap,
			       ",\"categories\":%d,\"parameters\":{\"alpha\":{\"id\":\"alpha\",\"type\":\"parameter\",\"value\":%.17g,\"lower\":0,\"upper\":\"infinity\"}}",
			       sp->categories, sp->alpha);
		if (sp->pinv >= 0)
			append(&js, &len, &cap, ",\"proportions\":{\"id\":\"pinv\",\"type\":\"Simplex\",\"values\":[%.17g,%.17g]}", sp->pinv, 1.0 - sp->pinv);
		append(&js, &len, &cap, "}");
	}
	if (sp->mu >= 0)
		append(&js, &len, &cap, ",\"mu\":{\"id\":\"mu\",\"type\":\"parameter\",\"value\":%.17g,\"lower\":0,\"upper\":\"infinity\"}", sp->mu);
	append(&js, &len, &cap, "}");
	if (!codon) {
		append(&js, &len, &cap, ",\"substitutionmodel\":{\"id\":\"sm\",\"type\":\"substitutionmodel\",\"model\":\"%s\",\"datatype\":\"%s\"", sp->model, sp->datatype);
		if (sp->nfreqs > 0) {
			append(&js, &len, &cap, ",\"frequencies\":{\"id\":\"freqs\",\"type\":\"Simplex\",\"values\":[");
			for (int i = 0; i < sp->nfreqs; i++) append(&js, &len, &cap, "%s%.17g", i ? "," : "", sp->freqs[i]);
			append(&js, &len, &cap, "]}");
		}
		if (!strcmp(sp->model, "gtr")) {
			const char *nm[5] = {"ac", "ag", "at", "cg", "ct"};
			append(&js, &len, &cap, ",\"rates\":{");
			for (int i = 0; i < 5; i++)
				append(&js, &len, &cap, "%s\"%s\":{\"id\":\"%s\",\"type\":\"parameter\",\"value\":%.17g,\"lower\":0,\"upper\":\"infinity\"}", i ? "," : "", nm[i], nm[i], sp->rates[i]);
			append(&js, &len, &cap, "}");
		} else if (!strcmp(sp->model, "hky")) {
			append(&js, &len, &cap, ",\"rates\":{\"kappa\":{\"id\":\"kappa\",\"type\":\"parameter\",\"value\":%.17g,\"lower\":0,\"upper\":\"infinity\"}}", sp->rates[0]);
		}
		append(&js, &len, &cap, "}");
	}
	append(&js, &len, &cap, "}");

	if (!codon) {
		b.json = create_json_tree(js);
		b.mlike = new_TreeLikelihoodModel_from_json(b.json, b.hash);
	} else {
		/* SubstitutionModel_factory has empty bodies for MG94 (substmodel.c:1526-1544): use the C constructors. */
		int S = b.patterns->nstate;
		Simplex *fs = new_Simplex("freqs", S);
		if (sp->nfreqs == S) fs->set_values(fs, sp->freqs);
		Model *mfs = new_SimplexModel("freqs", fs);
		SubstitutionModel *m = new_MG94_with_values(fs, sp->rates[1], sp->rates[2], sp->rates[0], 0);
		Model *mm = new_SubstitutionModel2("sm", m, mfs, NULL);
		mfs->free(mfs);
		/* site model through JSON (no substitution model inside) */
		char sjs[1024];
		if (sp->categories > 1)
			snprintf(sjs, sizeof sjs,
			         "{\"id\":\"sitemodel\",\"type\":\"sitemodel\",\"distribution\":{\"distribution\":\"gamma\",\"categories\":%d,\"parameters\":{\"alpha\":{\"id\":\"alpha\",\"type\":\"parameter\",\"value\":%.17g,\"lower\":0,\"upper\":\"infinity\"}}}}",
			         sp->categories, sp->alpha);
		else
			snprintf(sjs, sizeof sjs, "{\"id\":\"sitemodel\",\"type\":\"sitemodel\"}");
		b.json = create_json_tree(sjs);
		Model *msm = new_SiteModel_from_json(b.json, b.hash);
		SingleTreeLikelihood *tlk = new_SingleTreeLikelihood((Tree *)b.mtree->obj, m, (SiteModel *)msm->obj, b.patterns, NULL, sp->tipstates);
		b.patterns->ref_count++;
		b.mlike = new_TreeLikelihoodModel("treelikelihood", tlk, b.mtree, mm, msm, NULL);
		tlk->include_jacobian = false; /* physher.cpp:627: the C constructor leaves it unset */
		mm->free(mm);
		msm->free(msm);
	}
	free(js);
	SingleTreeLikelihood *tlk = b.mlike->obj;
	if (sp->generic_kernels || codon) {
		/* generic-state kernels (treelikelihoodX.c); the codon dispatcher is stale (SURVEY 8a notes) */
		SingleTreeLikelihood_enable_SSE(tlk, false);
		tlk->update_partials = update_partials_general;
		tlk->integrate_partials = integrate_partials_general;
		tlk->node_log_likelihoods = node_log_likelihoods_general;
		tlk->calculate_per_cat_partials = calculate_branch_partials;
	}
	if (sp->rescale) SingleTreeLikelihood_use_rescaling(tlk, true);
	for (int i = 0; i < ntaxa; i++) free(taxa[i]);
	free(taxa);
	return b;
}

/* The assembly of src/phycpp/physher.cpp:53-75 (data type), :321-350 (substitution model), :459-497 (site model)
 * and :594-629 (attribute pattern + likelihood), driven from an attr-spec file. */
static built_t build_from_attr_spec(const char *path) {
	built_t b;
	memset(&b, 0, sizeof(b));
	FILE *f = fopen(path, "r");
	if (!f) { fprintf(stderr, "cannot open spec %s\n", path); exit(2); }
	char key[64], val[8192], traits[1024] = "", newick_path[1024] = "";
	char *states[256];
	int nstates = 0, ncat = 1, normalize = 1, nrates = 0, nfreqs = 0, nstruct = 0;
	double alpha = 0.5, rates[512], freqs[256], structure_d[4096];
	char *amb[64];
	int namb = 0;
	while (fscanf(f, "%63s %8191s", key, val) == 2) {
		if (!strcmp(key, "states")) {
			char *save;
			for (char *tok = strtok_r(val, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) states[nstates++] = strdup(tok);
		} else if (!strcmp(key, "ambiguity")) amb[namb++] = strdup(val);
		else if (!strcmp(key, "traits")) strcpy(traits, val);
		else if (!strcmp(key, "newick")) strcpy(newick_path, val);
		else if (!strcmp(key, "structure")) nstruct = parse_list(val, structure_d, 4096);
		else if (!strcmp(key, "rates")) nrates = parse_list(val, rates, 512);
		else if (!strcmp(key, "freqs")) nfreqs = parse_list(val, freqs, 256);
		else if (!strcmp(key, "normalize")) normalize = atoi(val);
		else if (!strcmp(key, "categories")) ncat = atoi(val);
		else if (!strcmp(key, "alpha")) alpha = atof(val);
		else { fprintf(stderr, "unknown attr-spec key %s\n", key); exit(2); }
	}
	fclose(f);
	/* structure: S(S-1) entries = upper triangle then lower triangle, both row by row (gensubst.c:60-79).  The packed
	 * S(S-1)/2 form selects _reversible_update_Q, which indexes it as a full matrix (gensubst.c:130-151: out of bounds). */
	if (nfreqs != nstates || nstruct != nstates * (nstates - 1)) { fprintf(stderr, "attr spec: freqs/structure do not match the state count\n"); exit(2); }

	DataType *dt = new_GenericDataType("general_datatype", nstates, (const char **)states);
	for (int a = 0; a < namb; a++) {
		char *eq = strchr(amb[a], '=');
		*eq = 0;
		const char *members[256];
		int nm = 0;
		char *save;
		for (char *tok = strtok_r(eq + 1, "|", &save); tok; tok = strtok_r(NULL, "|", &save)) members[nm++] = tok;
		GenericDataType_add_ambiguity(dt, amb[a], nm, members);
	}

	char *taxa[4096], *values[4096];
	int ntaxa = 0;
	f = fopen(traits, "r");
	if (!f) { fprintf(stderr, "cannot open %s\n", traits); exit(2); }
	char a1[512], a2[512];
	while (fscanf(f, "%511s %511s", a1, a2) == 2) {
		taxa[ntaxa] = strdup(a1);
		values[ntaxa++] = strdup(a2);
	}
	fclose(f);
	b.patterns = new_AttributePattern(dt, (const char **)taxa, (const char **)values, ntaxa);

	char *newick = slurp(newick_path);
	b.mtree = new_TreeModel_from_newick(newick, taxa, NULL);
	free(newick);

	Parameters *rp = new_Parameters(nrates);
	for (int i = 0; i < nrates; i++) Parameters_move(rp, new_Parameter("subst_rates", rates[i], new_Constraint(0, INFINITY)));
	Simplex *fs = new_Simplex_with_values("subst_frequency_simplex", freqs, nfreqs);
	Model *mfs = new_SimplexModel("subst_frequencies", fs);
	unsigned structure[4096];
	for (int i = 0; i < nstruct; i++) structure[i] = (unsigned)structure_d[i];
	DiscreteParameter *dp = new_DiscreteParameter_with_values(structure, nstruct);
	Model *mdp = new_DiscreteParameterModel("structure", dp);
	SubstitutionModel *m = new_GeneralModel_with_parameters(dt, (DiscreteParameter *)mdp->obj, rp, fs, -1, normalize);
	Model *mm = new_SubstitutionModel3("substmodel", m, mfs, NULL, mdp);
	free_Parameters(rp);
	mfs->free(mfs);

	b.hash = new_Hashtable_string(10);
	hashtable_set_key_ownership(b.hash, false);
	hashtable_set_value_ownership(b.hash, false);
	char sjs[1024];
	if (ncat > 1)
		snprintf(sjs, sizeof sjs,
		         "{\"id\":\"sitemodel\",\"type\":\"sitemodel\",\"distribution\":{\"distribution\":\"gamma\",\"categories\":%d,\"parameters\":{\"alpha\":{\"id\":\"alpha\",\"type\":\"parameter\",\"value\":%.17g,\"lower\":0,\"upper\":\"infinity\"}}}}",
		         ncat, alpha);
	else
		snprintf(sjs, sizeof sjs, "{\"id\":\"sitemodel\",\"type\":\"sitemodel\"}");
	b.json = create_json_tree(sjs);
	Model *msm = new_SiteModel_from_json(b.json, b.hash);
	SingleTreeLikelihood *tlk = new_SingleTreeLikelihood((Tree *)b.mtree->obj, m, (SiteModel *)msm->obj, b.patterns, NULL, false);
	b.mlike = new_TreeLikelihoodModel("treelike", tlk, b.mtree, mm, msm, NULL);
	tlk->include_jacobian = false; /* physher.cpp:627: the C constructor leaves it unset */
	/* the scalar generic-state kernels: the SSE "even" variants are not valid for odd state counts (treelikelihood.c:1150-1152) */
	SingleTreeLikelihood_enable_SSE(tlk, false);
	tlk->update_partials = update_partials_general;
	tlk->integrate_partials = integrate_partials_general;
	tlk->node_log_likelihoods = node_log_likelihoods_general;
	tlk->calculate_per_cat_partials = calculate_branch_partials;
	return b;
}

static void jnum(FILE *o, double v);

/* With integration/physher_device.c in front of libphyc (LD_PRELOAD) the model may be evaluated on the GPU: the CPU-side
 * partial arrays are then never filled and are not dumped. */
static int on_device(SingleTreeLikelihood *tlk) {
	int (*f)(const SingleTreeLikelihood *) = (int (*)(const SingleTreeLikelihood *))dlsym(RTLD_DEFAULT, "SingleTreeLikelihood_on_device");
	return f ? f(tlk) : 0;
}
static int device_rescaling(SingleTreeLikelihood *tlk) {
	int (*f)(const SingleTreeLikelihood *) = (int (*)(const SingleTreeLikelihood *))dlsym(RTLD_DEFAULT, "SingleTreeLikelihood_device_is_rescaling");
	return f ? f(tlk) : 0;
}

static void jarr(FILE *o, const char *key, const double *v, size_t n, bool comma) {
	fprintf(o, "\"%s\":[", key);
	for (size_t i = 0; i < n; i++) {
		if (i) fprintf(o, ",");
		jnum(o, v[i]);
	}
	fprintf(o, "]%s\n", comma ? "," : "");
}

static void jnum(FILE *o, double v) {
	if (isnan(v)) fprintf(o, "NaN");
	else if (isinf(v)) fprintf(o, v > 0 ? "Infinity" : "-Infinity");
	else fprintf(o, "%.17g", v);
}

static void dump_common(FILE *o, Model *mlike) {
	SingleTreeLikelihood *tlk = mlike->obj;
	Tree *tree = tlk->tree;
	SitePattern *sp = tlk->sp;
	int N = Tree_node_count(tree), T = Tree_tip_count(tree), P = sp->count, S = tlk->m->nstate, C = tlk->cat_count;

	fprintf(o, "\"tip_count\":%d,\"node_count\":%d,\"pattern_count\":%d,\"state_count\":%d,\"category_count\":%d,\n", T, N, P, S, C);
	fprintf(o, "\"taxa\":[");
	for (int i = 0; i < sp->size; i++) fprintf(o, "%s\"%s\"", i ? "," : "", sp->names[i]);
	fprintf(o, "],\n");
	jarr(o, "weights", sp->weights, P, true);
	fprintf(o, "\"patterns\":[");
	for (int i = 0; i < sp->size; i++) {
		fprintf(o, "%s[", i ? "," : "");
		for (int k = 0; k < P; k++) fprintf(o, "%s%d", k ? "," : "", (int)sp->patterns[i][k]);
		fprintf(o, "]");
	}
	fprintf(o, "],\n");

	/* first evaluation: lnL (also fills matrices, partials, pattern_lk) */
	double lnl = mlike->logP(mlike);
	fprintf(o, "\"lnl\":");
	jnum(o, lnl);
	fprintf(o, ",\n\"rescaled\":%s,\n", (tlk->scale || device_rescaling(tlk)) ? "true" : "false");
	jarr(o, "pattern_lk", tlk->pattern_lk, P, true);

	fprintf(o, "\"nodes\":[");
	for (int i = 0; i < N; i++) {
		Node *n = Tree_node(tree, i);
		fprintf(o, "%s{\"id\":%d,\"class_id\":%d,\"name\":\"%s\",\"left\":%d,\"right\":%d,\"parent\":%d,\"distance\":%.17g,\"mapping\":%d}", i ? "," : "", n->id,
		        n->class_id, n->name ? n->name : "", n->left ? n->left->id : -1, n->right ? n->right->id : -1, n->parent ? n->parent->id : -1,
		        Node_distance(n), tlk->mapping[n->id]);
	}
	fprintf(o, "],\n\"root\":%d,\n", Tree_root(tree)->id);

	tlk->sm->update(tlk->sm);
	double *rates = malloc(sizeof(double) * C);
	for (int c = 0; c < C; c++) rates[c] = tlk->sm->get_rate(tlk->sm, c);
	jarr(o, "cat_rates", rates, C, true);
	free(rates);
	jarr(o, "cat_proportions", tlk->sm->get_proportions(tlk->sm), C, true);
	jarr(o, "cat_rates_without_mu", tlk->sm->cat_rates, C, true);
	fprintf(o, "\"site_rate_parameters\":%d,\"site_has_pinv\":%s,\"site_has_mu\":%s,\n", (int)Parameters_count(tlk->sm->rates),
	        tlk->sm->proportions ? "true" : "false", tlk->sm->mu ? "true" : "false");
	jarr(o, "frequencies", tlk->get_root_frequencies(tlk), S, true);
	jarr(o, "eval", tlk->m->eigendcmp->eval, S, true);
	{
		double *tmp = malloc(sizeof(double) * S * S);
		for (int i = 0; i < S; i++)
			for (int j = 0; j < S; j++) tmp[i * S + j] = tlk->m->eigendcmp->evec[i][j];
		jarr(o, "evec", tmp, (size_t)S * S, true);
		for (int i = 0; i < S; i++)
			for (int j = 0; j < S; j++) tmp[i * S + j] = tlk->m->eigendcmp->Invevec[i][j];
		jarr(o, "ivec", tmp, (size_t)S * S, true);
		if (tlk->m->Q) {
			for (int i = 0; i < S; i++)
				for (int j = 0; j < S; j++) tmp[i * S + j] = tlk->m->Q[i][j];
			jarr(o, "Q", tmp, (size_t)S * S, true);
		}
		free(tmp);
	}
	/* P(t) of three non-root nodes: node 0 (a tip), first internal, last non-root.  Fresh p_t calls (never transposed). */
	{
		int pick[3] = {0, T, N - 2};
		double *mat = malloc(sizeof(double) * S * S * 2);
		fprintf(o, "\"pt_nodes\":[%d,%d,%d],\n", pick[0], pick[1], pick[2]);
		fprintf(o, "\"pt\":[");
		for (int q = 0; q < 3; q++) {
			Node *n = Tree_node(tree, pick[q]);
			fprintf(o, "%s[", q ? "," : "");
			for (int c = 0; c < C; c++) {
				tlk->m->p_t(tlk->m, Node_distance(n) * tlk->sm->get_rate(tlk->sm, c), mat);
				for (int i = 0; i < S * S; i++) fprintf(o, "%s%.17g", (c || i) ? "," : "", mat[i]);
			}
			fprintf(o, "]");
		}
		fprintf(o, "],\n\"dpt\":[");
		for (int q = 0; q < 3; q++) {
			Node *n = Tree_node(tree, pick[q]);
			fprintf(o, "%s[", q ? "," : "");
			for (int c = 0; c < C; c++) {
				tlk->m->dp_dt(tlk->m, Node_distance(n) * tlk->sm->get_rate(tlk->sm, c), mat);
				for (int i = 0; i < S * S; i++) fprintf(o, "%s%.17g", (c || i) ? "," : "", mat[i]);
			}
			fprintf(o, "]");
		}
		fprintf(o, "],\n");
		free(mat);
	}
	/* lower partials of the root and of the first internal node, reference layout [C][P][S] */
	if (!on_device(tlk)) {
		int ids[2] = {T, Tree_root(tree)->id};
		const char *nm[2] = {"partials_first_internal", "partials_root"};
		for (int q = 0; q < 2; q++) jarr(o, nm[q], tlk->partials[tlk->current_partials_indexes[ids[q]]][ids[q]], (size_t)C * P * S, true);
		if (tlk->scale) jarr(o, "scaling_root", tlk->scaling_factors[tlk->current_partials_indexes[ids[1]]][ids[1]], P, true);
	}
}

static void dump_gradients_unrooted(FILE *o, Model *mlike) {
	SingleTreeLikelihood *tlk = mlike->obj;
	int N = Tree_node_count(tlk->tree);
	/* branch-length gradient only (the headline metric): flag TREE_MODEL */
	size_t len = TreeLikelihood_initialize_gradient(mlike, TREELIKELIHOOD_FLAG_TREE_MODEL);
	SingleTreeLikelihood_update_all_nodes(tlk);
	double *g = TreeLikelihood_gradient(mlike);
	jarr(o, "gradient_tree", g, len, true);
	/* upper partials of the first internal node and of tip 0 (valid after the gradient call) */
	if (!on_device(tlk)) {
		int T = Tree_tip_count(tlk->tree);
		size_t sz = (size_t)tlk->cat_count * tlk->sp->count * tlk->m->nstate;
		int ids[2] = {0, T};
		const char *nm[2] = {"upper_tip0", "upper_first_internal"};
		for (int q = 0; q < 2; q++) {
			int idx = tlk->upper_partial_indexes[ids[q]];
			jarr(o, nm[q], tlk->partials[tlk->current_partials_indexes[idx]][idx], sz, true);
		}
	}
	/* tree + site model + substitution model ("next"-tier rows G2).  flags==0 is NOT used: with it
	 * TreeLikelihood_initialize_gradient under-counts the substitution block (treelikelihood.c:248-249
	 * are evaluated before :259) and TreeLikelihood_calculate_gradient overruns tlk->gradient. */
	{
		int flags = TREELIKELIHOOD_FLAG_TREE_MODEL;
		if (tlk->sm->proportions != NULL || Parameters_count(tlk->sm->rates) > 0 || tlk->sm->mu != NULL) flags |= TREELIKELIHOOD_FLAG_SITE_MODEL;
		if (tlk->m->dPdp != NULL) flags |= TREELIKELIHOOD_FLAG_SUBSTITUTION_MODEL;
		len = TreeLikelihood_initialize_gradient(mlike, flags);
		SingleTreeLikelihood_update_all_nodes(tlk);
		g = TreeLikelihood_gradient(mlike);
		jarr(o, "gradient_all", g, len, true);
		fprintf(o, "\"gradient_all_flags\":%d,\"gradient_all_tree_len\":%d,\n", flags, N);
	}
}

/* peak resident memory of THIS process image in kB (VmHWM of /proc/self/status: unlike wait4's ru_maxrss it does not
 * include what the parent held when it forked) */
static long peak_rss_kb(void) {
	FILE *f = fopen("/proc/self/status", "r");
	if (!f) return -1;
	char line[256];
	long kb = -1;
	while (fgets(line, sizeof line, f))
		if (!strncmp(line, "VmHWM:", 6)) kb = atol(line + 6);
	fclose(f);
	return kb;
}

static double now_ms(void) {
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC_RAW, &t);
	return t.tv_sec * 1000. + t.tv_nsec / 1e6;
}

int main(int argc, char **argv) {
	if (argc < 4) {
		fprintf(stderr, "usage: %s dump|json|bench ...\n", argv[0]);
		return 2;
	}
	if (!strcmp(argv[1], "dump")) {
		spec_t sp;
		read_spec(argv[2], &sp);
		built_t b = build_from_spec(&sp);
		FILE *o = fopen(argv[3], "w");
		fprintf(o, "{\n");
		dump_common(o, b.mlike);
		dump_gradients_unrooted(o, b.mlike);
		fprintf(o, "\"source\":\"physher reference (libphyc no-GSL build) via oracle/ref_driver.c\"\n}\n");
		fclose(o);
		return 0;
	}
	if (!strcmp(argv[1], "branch")) {
		/* The optimiser's fast path: lnL and its first two derivatives with respect to ONE branch length at trial
		 * values, by the reference's own upper-partial protocol (_singleTreeLikelihood_d2logP, treelikelihood.c:469-530):
		 * full evaluation at the trial length, update_upper_partials, calculate_dldt_uppper, d2lnldt2_uppper. */
		spec_t sp;
		read_spec(argv[2], &sp);
		built_t b = build_from_spec(&sp);
		SingleTreeLikelihood *tlk = b.mlike->obj;
		Tree *tree = tlk->tree;
		const int N = Tree_node_count(tree), T = Tree_tip_count(tree), P = tlk->sp->count;
		const int right_of_root = Node_id(Node_right(Tree_root(tree)));
		int pick[4] = {0, T / 2, T, N - 2};
		const double factor[3] = {1.0, 0.4, 2.5};
		FILE *o = fopen(argv[3], "w");
		fprintf(o, "{\"trials\":[");
		int first = 1;
		for (int q = 0; q < 4; q++) {
			int id = pick[q];
			while (id == right_of_root || id == Node_id(Tree_root(tree))) id--;
			Node *node = Tree_node(tree, id);
			const double base = Node_distance(node);
			for (int f = 0; f < 3; f++) {
				const double t = base * factor[f];
				Node_set_distance(node, t);
				SingleTreeLikelihood_update_all_nodes(tlk);
				const double lnl = b.mlike->logP(b.mlike);
				update_upper_partials(tlk, Tree_root(tree), false);
				double *pl = tlk->pattern_lk + P, *pdl = tlk->pattern_lk + 2 * P;
				for (int k = 0; k < P; k++) pl[k] = exp(tlk->pattern_lk[k]);
				calculate_dldt_uppper(tlk, node, pdl);
				double d1 = 0;
				for (int k = 0; k < P; k++) d1 += tlk->sp->weights[k] * pdl[k] / pl[k];
				const double d2 = d2lnldt2_uppper(tlk, node, pl, pdl);
				fprintf(o, "%s{\"node\":%d,\"length\":%.17g,\"lnl\":%.17g,\"d1\":%.17g,\"d2\":%.17g}", first ? "" : ",", id, t, lnl, d1, d2);
				first = 0;
			}
			Node_set_distance(node, base);
		}
		fprintf(o, "],\"rescaled\":%s,\"source\":\"physher reference via oracle/ref_driver.c branch mode\"}\n", tlk->scale ? "true" : "false");
		fclose(o);
		return 0;
	}
	if (!strcmp(argv[1], "brent")) {
		/* The call pattern of serial_brent_optimize_tree (optimizer.c:112-153) without the line search itself: use_upper on, the
		 * branches in post-order, three trial lengths per branch through Node_set_distance + logP (single-branch evaluations
		 * from the partials that meet on the branch, _calculate :1558-1606), the last trial kept; then use_upper off, one plain
		 * logP and the TREE_MODEL gradient -- which must see every accepted length. */
		spec_t sp;
		read_spec(argv[2], &sp);
		built_t b = build_from_spec(&sp);
		SingleTreeLikelihood *tlk = b.mlike->obj;
		Tree *tree = tlk->tree;
		const int N = Tree_node_count(tree);
		Node **nodes = Tree_get_nodes(tree, POSTORDER);
		const double factor[3] = {0.7, 1.6, 1.15};
		FILE *o = fopen(argv[3], "w");
		fprintf(o, "{\"lnl_start\":%.17g,\n\"trials\":[", b.mlike->logP(b.mlike));
		tlk->node_upper = NULL;
		tlk->use_upper = true;
		tlk->update_upper = true;
		if (Node_distance(Tree_root(tree)->right) != 0) {
			double tot = Node_distance(Tree_root(tree)->right) + Node_distance(Tree_root(tree)->left);
			Node_set_distance(Tree_root(tree)->right, 0);
			Node_set_distance(Tree_root(tree)->left, tot);
		}
		SingleTreeLikelihood_update_uppers(tlk);
		int first = 1;
		for (int i = 0; i < N; i++) {
			Node *node = nodes[i];
			if (Node_isroot(node) || (Node_isroot(Node_parent(node)) && Node_right(Node_parent(node)) == node)) continue;
			if (tlk->node_upper == NULL) tlk->node_upper = node;
			const double base = Node_distance(node);
			double l0 = b.mlike->logP(b.mlike); /* Brent's first point: the current value, nothing changed */
			fprintf(o, "%s{\"node\":%d,\"length\":%.17g,\"lnl\":%.17g}", first ? "" : ",", Node_id(node), base, l0);
			first = 0;
			for (int f = 0; f < 3; f++) {
				Node_set_distance(node, base * factor[f]);
				fprintf(o, ",{\"node\":%d,\"length\":%.17g,\"lnl\":%.17g}", Node_id(node), base * factor[f], b.mlike->logP(b.mlike));
			}
		}
		tlk->use_upper = false;
		fprintf(o, "],\n\"lnl_end\":%.17g,\n", b.mlike->logP(b.mlike));
		SingleTreeLikelihood_update_all_nodes(tlk);
		fprintf(o, "\"lnl_end_recomputed\":%.17g,\n", b.mlike->logP(b.mlike));
		size_t len = TreeLikelihood_initialize_gradient(b.mlike, TREELIKELIHOOD_FLAG_TREE_MODEL);
		double *g = TreeLikelihood_gradient(b.mlike);
		jarr(o, "gradient_tree_end", g, len, true);
		fprintf(o, "\"source\":\"physher reference via oracle/ref_driver.c brent mode\"}\n");
		fclose(o);
		return 0;
	}
	if (!strcmp(argv[1], "attr")) {
		built_t b = build_from_attr_spec(argv[2]);
		FILE *o = fopen(argv[3], "w");
		fprintf(o, "{\n");
		dump_common(o, b.mlike);
		dump_gradients_unrooted(o, b.mlike);
		fprintf(o, "\"source\":\"physher reference via oracle/ref_driver.c attr mode\"\n}\n");
		fclose(o);
		return 0;
	}
	if (!strcmp(argv[1], "json")) {
		/* a reference-format JSON document whose first child is a treelikelihood node (tests/data/jc69-time.json) */
		Hashtable *hash = new_Hashtable_string(10);
		hashtable_set_key_ownership(hash, false);
		hashtable_set_value_ownership(hash, false);
		char *content = load_file(argv[2]);
		json_node *json = create_json_tree(content);
		free(content);
		Model *model = new_TreeLikelihoodModel_from_json(json->children[0], hash);
		SingleTreeLikelihood *tlk = model->obj;
		Tree_update_heights(tlk->tree);
		FILE *o = fopen(argv[3], "w");
		fprintf(o, "{\n");
		for (int jac = 0; jac < 2; jac++) {
			tlk->include_jacobian = jac;
			SingleTreeLikelihood_update_all_nodes(tlk);
			double lnl = model->logP(model);
			fprintf(o, "\"lnl_jacobian%d\":%.17g,\n", jac, lnl);
			size_t len = TreeLikelihood_initialize_gradient(model, TREELIKELIHOOD_FLAG_TREE_MODEL | TREELIKELIHOOD_FLAG_BRANCH_MODEL);
			SingleTreeLikelihood_update_all_nodes(tlk);
			double *g = TreeLikelihood_gradient(model);
			char key[64];
			snprintf(key, sizeof key, "gradient_tree_clock_jacobian%d", jac);
			jarr(o, key, g, len, true);
		}
		tlk->include_jacobian = false;
		SingleTreeLikelihood_update_all_nodes(tlk);
		{
			/* every block the model can differentiate: [tree (ratios / root height)][site model][clock][substitution model] */
			int flags = TREELIKELIHOOD_FLAG_TREE_MODEL | TREELIKELIHOOD_FLAG_BRANCH_MODEL;
			if (tlk->sm->proportions != NULL || Parameters_count(tlk->sm->rates) > 0 || tlk->sm->mu != NULL) flags |= TREELIKELIHOOD_FLAG_SITE_MODEL;
			if (tlk->m->dPdp != NULL) flags |= TREELIKELIHOOD_FLAG_SUBSTITUTION_MODEL;
			size_t len = TreeLikelihood_initialize_gradient(model, flags);
			SingleTreeLikelihood_update_all_nodes(tlk);
			double *g = TreeLikelihood_gradient(model);
			jarr(o, "gradient_all_time", g, len, true);
			fprintf(o, "\"gradient_all_time_flags\":%d,\n", flags);
			SingleTreeLikelihood_update_all_nodes(tlk);
		}
		{
			Tree *tree = tlk->tree;
			int N = Tree_node_count(tree);
			double *h = malloc(sizeof(double) * N), *bl = malloc(sizeof(double) * N);
			for (int i = 0; i < N; i++) {
				Node *n = Tree_node(tree, i);
				h[i] = Node_height(n);
				bl[i] = Node_isroot(n) ? 0 : tlk->bm->get(tlk->bm, n) * Node_time_elapsed(n);
			}
			jarr(o, "heights", h, N, true);
			jarr(o, "branch_lengths", bl, N, true);
			free(h);
			free(bl);
		}
		dump_common(o, model);
		fprintf(o, "\"source\":\"physher reference via oracle/ref_driver.c json mode\"\n}\n");
		fclose(o);
		return 0;
	}
	if (!strcmp(argv[1], "bench")) {
		spec_t sp;
		read_spec(argv[2], &sp);
		int iters = atoi(argv[3]);
		int warm = argc > 4 ? atoi(argv[4]) : 1;
		built_t b = build_from_spec(&sp);
		SingleTreeLikelihood *tlk = b.mlike->obj;
		double lnl = 0;
		for (int i = 0; i < warm; i++) {
			SingleTreeLikelihood_update_all_nodes(tlk);
			tlk->m->need_update = true;
			lnl = b.mlike->logP(b.mlike);
		}
		double t0 = now_ms();
		for (int i = 0; i < iters; i++) {
			SingleTreeLikelihood_update_all_nodes(tlk);
			tlk->m->need_update = true;
			lnl = b.mlike->logP(b.mlike);
		}
		double t1 = now_ms();
		TreeLikelihood_initialize_gradient(b.mlike, TREELIKELIHOOD_FLAG_TREE_MODEL);
		for (int i = 0; i < warm; i++) {
			SingleTreeLikelihood_update_all_nodes(tlk);
			tlk->m->need_update = true;
			TreeLikelihood_gradient(b.mlike);
		}
		double t2 = now_ms();
		for (int i = 0; i < iters; i++) {
			SingleTreeLikelihood_update_all_nodes(tlk);
			tlk->m->need_update = true;
			TreeLikelihood_gradient(b.mlike);
		}
		double t3 = now_ms();
		printf("{\"lnl\":%.17g,\"iters\":%d,\"lnl_ms_per_eval\":%.6f,\"grad_ms_per_eval\":%.6f,\"patterns\":%d,\"taxa\":%d,\"rescaled\":%s,\"peak_rss_kb\":%ld}\n", lnl,
		       iters, (t1 - t0) / iters, (t3 - t2) / iters, tlk->sp->count, Tree_tip_count(tlk->tree), tlk->scale ? "true" : "false", peak_rss_kb());
		if (argc > 5) {
			/* what the timed protocol computed, for the caller to hold its own numbers against: the gradient vector of one more
			 * TreeLikelihood_gradient call (TREE_MODEL flag, entry = node id), the per-pattern lnL, and the compressed patterns
			 * (one string per taxon, character '0' + state code) with their weights */
			Tree *tree = tlk->tree;
			SitePattern *pat = tlk->sp;
			const int N = Tree_node_count(tree);
			SingleTreeLikelihood_update_all_nodes(tlk);
			tlk->m->need_update = true;
			double *g = TreeLikelihood_gradient(b.mlike);
			FILE *o = fopen(argv[5], "w");
			if (!o) { fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
			fprintf(o, "{\n");
			jarr(o, "gradient", g, N, true);
			SingleTreeLikelihood_update_all_nodes(tlk);
			tlk->m->need_update = true;
			lnl = b.mlike->logP(b.mlike);
			fprintf(o, "\"lnl\":%.17g,\n", lnl);
			jarr(o, "pattern_lk", tlk->pattern_lk, pat->count, true);
			jarr(o, "weights", pat->weights, pat->count, true);
			fprintf(o, "\"taxa\":[");
			for (int i = 0; i < pat->size; i++) fprintf(o, "%s\"%s\"", i ? "," : "", pat->names[i]);
			fprintf(o, "],\n\"patterns\":[");
			for (int i = 0; i < pat->size; i++) {
				fprintf(o, "%s\"", i ? "," : "");
				for (int k = 0; k < pat->count; k++) fputc('0' + (int)pat->patterns[i][k], o);
				fprintf(o, "\"");
			}
			fprintf(o, "],\n\"nodes\":[");
			for (int i = 0; i < N; i++) {
				Node *n = Tree_node(tree, i);
				fprintf(o, "%s{\"id\":%d,\"name\":\"%s\",\"left\":%d,\"right\":%d,\"distance\":%.17g}", i ? "," : "", n->id, n->name ? n->name : "",
				        n->left ? n->left->id : -1, n->right ? n->right->id : -1, Node_distance(n));
			}
			fprintf(o, "],\n\"root\":%d\n}\n", Tree_root(tree)->id);
			fclose(o);
		}
		return 0;
	}
	if (!strcmp(argv[1], "toggle")) {
		/* device -> CPU on a live object: lnL and the TREE_MODEL gradient with the binding enabled (as the environment asks), then
		 * SingleTreeLikelihood_disable_device and the same two on the reference's own kernels -- an object built without host
		 * partial arrays (integration/physher_device.c, "lean") has to get the reference's storage back.  Prints both. */
		spec_t sp;
		read_spec(argv[2], &sp);
		built_t b = build_from_spec(&sp);
		SingleTreeLikelihood *tlk = b.mlike->obj;
		const int N = Tree_node_count(tlk->tree);
		FILE *o = fopen(argv[3], "w");
		fprintf(o, "{\n\"on_device_before\":%d,\n", on_device(tlk));
		for (int pass = 0; pass < 2; pass++) {
			SingleTreeLikelihood_update_all_nodes(tlk);
			double lnl = b.mlike->logP(b.mlike);
			TreeLikelihood_initialize_gradient(b.mlike, TREELIKELIHOOD_FLAG_TREE_MODEL);
			SingleTreeLikelihood_update_all_nodes(tlk);
			double *g = TreeLikelihood_gradient(b.mlike);
			fprintf(o, "\"lnl%d\":%.17g,\n", pass, lnl);
			char key[32];
			snprintf(key, sizeof key, "gradient%d", pass);
			jarr(o, key, g, N, true);
			if (pass == 0) {
				void (*off)(SingleTreeLikelihood *) = (void (*)(SingleTreeLikelihood *))dlsym(RTLD_DEFAULT, "SingleTreeLikelihood_disable_device");
				if (off) off(tlk);
			}
		}
		fprintf(o, "\"on_device_after\":%d\n}\n", on_device(tlk));
		fclose(o);
		return 0;
	}
	fprintf(stderr, "unknown mode %s\n", argv[1]);
	return 2;
}
