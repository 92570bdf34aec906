// phyamd_host.hpp -- host-side model code for the tree-likelihood hot path (C++17, no GPU code here).
//
// What stays on the host in physher stays on the host here: Newick parsing and the
// node-id convention, sequence encoding and site-pattern compression, substitution-model eigen systems,
// discrete-rate site models, the node-height (ratio) transform and the O(N) gradient epilogue.  Everything
// O(patterns x nodes) goes through the C ABI in include/physher_amd.h.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace phyamd {

struct Error : std::runtime_error {
	using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------------------------------------
// Tree (reference: src/phyc/tree.c, node.c, treetransform.c)
// ---------------------------------------------------------------------------------------------
struct Tree {
	int tip_count = 0, node_count = 0, root = -1;
	// indexed by node id: tips 0..T-1 = index in the taxon list, internals T.. in post-order (tree.c:202-224)
	std::vector<int> left, right, parent, class_id;
	std::vector<std::string> name;
	std::vector<double> distance;  // branch above the node (unrooted mode)
	std::vector<double> height;    // time mode
	std::vector<int> postorder, preorder;  // node ids, left before right (tree.c:1772-1790)
	bool time_mode = false;

	// time trees: ratio/root-height reparameterisation (treetransform.c)
	std::vector<double> lowers;  // by node id: max tip height below the node
	std::vector<double> ratios;  // by class_id (internal nodes); the root's entry is its height
	bool reparameterized = false;

	bool is_leaf(int n) const { return left[n] < 0; }
	int sibling(int n) const { return left[parent[n]] == n ? right[parent[n]] : left[parent[n]]; }
};

// Newick -> binary tree with the reference's conventions (tree.c:577-788): polytomies are resolved by inserting
// zero-length (tip) or BL_MIN (clade) nodes to the right; ids from `taxa`.  contain_bl: lengths are floored at BL_MIN.
Tree parse_newick(const std::string &newick, const std::vector<std::string> &taxa, bool contain_bl);
// new_TreeModel_from_newick with dates == NULL (tree.c:1409-1445): unrooted, root->right folded into root->left
Tree make_unrooted_tree(const std::string &newick, const std::vector<std::string> &taxa);
// new_TimeTreeModel_from_newick (tree.c:1447-1462): tip heights from dates (init_dates2, tree.c:394-424),
// internal heights from distances (tree.c:498-515)
Tree make_time_tree(const std::string &newick, const std::vector<std::string> &taxa, const std::vector<double> &dates);
// TreeModel_set_transform(RATIO): lowers + ratios from the current heights (treetransform.c:248-262, tree.c:516-535)
void enable_ratio_transform(Tree &t);
void heights_from_ratios(Tree &t);  // tree_transform_update_heights (treetransform.c:224-238)
// d f / d(ratios, root height) given d f / d heights (both by class_id): Tree_node_transform_jvp
void ratio_transform_jvp(const Tree &t, const double *height_gradient, double *gradient);
double ratio_transform_log_jacobian(const Tree &t);                     // treetransform.c:214-222
void ratio_transform_log_jacobian_gradient(const Tree &t, double *gradient);  // += (treetransform.c:183-212)

// ---------------------------------------------------------------------------------------------
// Sequences and site patterns (reference: datatype.c, sitepattern.c, hashtable.c)
// ---------------------------------------------------------------------------------------------
enum class DataTypeKind { Nucleotide, AminoAcid, Codon, General };

struct DataType {
	DataTypeKind kind = DataTypeKind::Nucleotide;
	int state_count = 4;
	int symbol_length = 1;
	std::vector<std::string> states;  // General
	// General: named ambiguity sets (GenericDataType_add_ambiguity, datatype.c:243-262): code = state_count + index here,
	// "unknown" = state_count + ambiguities.size() (datatype.c:184-199)
	std::vector<std::pair<std::string, std::vector<int>>> ambiguities;
	int encode(const char *sym) const;                 // datatype.c:55-89, sitepattern.c:796-819
	int encode_string(const std::string &s) const;     // whole-string lookup (_encoding_string, datatype.c:184-199): attribute patterns
	void partial(int code, double *out) const;         // ambiguity mask / one-hot / all ones (datatype.h:26-66, datatype.c:212-240)
};

struct Patterns {
	int taxon_count = 0, pattern_count = 0, site_count = 0;
	std::vector<std::string> names;
	std::vector<uint8_t> states;  // [taxon][pattern]
	std::vector<double> weights;  // [pattern]
};

// new_SitePattern (sitepattern.c:186-251): de-duplicate columns, bit-exact in the reference's hashtable order
Patterns compress_patterns(const DataType &dt, const std::vector<std::string> &names, const std::vector<std::string> &sequences);
// the same result from the device (phyamd_compress_patterns: hashing, grouping and the table order by radix sorts); throws
// if the device path declines (hash collision) -- callers fall back to compress_patterns
Patterns compress_patterns_device(const DataType &dt, const std::vector<std::string> &names, const std::vector<std::string> &sequences);
// alignments with at least this many sites are compressed on the device by the likelihood wrapper
constexpr size_t kDeviceCompressionSites = 50000;

// ---------------------------------------------------------------------------------------------
// Substitution models (reference: substmodel.c, gtr.c, hky.c, jc69.c, gensubst.c, eigen.c)
// ---------------------------------------------------------------------------------------------
struct SubstModel {
	int S = 4;
	std::string name;
	std::vector<double> rates;        // model-specific (HKY: kappa; GTR: 5 relative or 6 simplex; General: by structure)
	std::vector<double> freqs;        // [S]
	std::vector<unsigned> structure;  // General: upper-triangle -> rate index (gensubst.c)
	bool normalize = true;
	// derived
	std::vector<double> Q, eval, evec, ivec;  // row-major
	double norm = 1.0;                        // normalising constant the unnormalised Q was divided by (1 if !normalize)
	bool dirty = true;
	void update();                                // build Q (normalised, rows sum to 0) and its eigen system
	void p_t(double t, double *P, bool derivative = false);  // host reference of M1/M2, used by tests
	// Differentiable parameters in the reference's order and its constrained-value convention (grad_wrt_reparam = false,
	// treelikelihood.c:296-305): the rates as stored, then every frequency as a free coordinate.
	int rate_parameter_count() const { return (name == "JC69" || name == "WAG" || name == "LG") ? 0 : (int)rates.size(); }
	// d(normalised Q)/d(parameter), [count][S][S]: _gtr_dQdp / _hky_dQdp / _general_dQdp (gtr.c:256-326, hky.c:493-541,
	// gensubst.c:216-279) in one rule: dQ^ = dR o pi + R o dpi (rows re-zeroed), dQ = (dQ^ - Q dnorm) / norm
	void rate_matrix_derivatives(bool want_rates, bool want_freqs, std::vector<double> &dQ);
};
void build_symmetric_rates(const SubstModel &m, std::vector<double> &R);  // exchangeabilities r_ij (i < j), row-major full matrix
// model names: JC69, HKY, GTR, GENERAL (above) and the fixed-exchangeability models of the 20- and 61-state configurations:
//   WAG, LG  (wag.c:23-36, lg.c:23-36): tabulated exchangeabilities, any frequencies (wag_frequencies() = WAG's own)
//   MG94     (mg94.c:62-138): universal code, rates = {kappa, alpha, beta}, 61 codon frequencies
const double *wag_frequencies();  // [20]

// ---------------------------------------------------------------------------------------------
// Site models (reference: sitemodel.c, gamma.c)
// ---------------------------------------------------------------------------------------------
enum class RateDistribution { Constant, Gamma, Weibull };
struct SiteModel {
	RateDistribution dist = RateDistribution::Constant;
	int cat_count = 1;  // includes the invariant category
	double shape = 1.0;
	bool has_pinv = false;
	double pinv = 0.0;
	bool has_mu = false;
	double mu = 1.0;
	std::vector<double> cat_rates, cat_props;  // without mu
	bool dirty = true;
	void update();
	double rate(int c) { update(); return cat_rates[c] * (has_mu ? mu : 1.0); }
	// ingrad[c] = sum over branches of cat_gradient[branch][c] * branch_length (x mu); for the +I term ingrad[0] is the
	// root-partial sum of phyamd_root_invariant_term
	double shape_gradient(const double *ingrad);  // _gamma_shape_derivative / _weibull_shape_derivative (sitemodel.c:258-308, 375-398)
	double pinv_gradient(const double *ingrad);   // _gamma_inv_derivative / _weibull_inv_derivative (sitemodel.c:310-341, 400-426)
	double quantile(double p, double shape_value) const;  // un-normalised class rate
};
double gamma_quantile(double p, double shape, double rate);  // lower-tail inverse CDF (gamma.c:55-58,194-228)
double reg_lower_gamma(double a, double x);                  // P(a, x)

}  // namespace phyamd
