// interfaces.cpp -- the phycpp-compatible wrapper classes (include/phycpp_amd/physher.hpp) over the host model
// code (phyamd_host.hpp) and the device engine's C ABI (include/physher_amd.h).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "phyamd_host.hpp"
#include "phycpp_amd/physher.hpp"
#include "physher_amd.h"

using phyamd::Error;

// ---------------------------------------------------------------------------------------------
// data types
// ---------------------------------------------------------------------------------------------
DataTypeInterface::DataTypeInterface() : dataType_(std::make_shared<phyamd::DataType>()) {}
DataTypeInterface::~DataTypeInterface() = default;

NucleotideDataTypeInterface::NucleotideDataTypeInterface() {
	dataType_->kind = phyamd::DataTypeKind::Nucleotide;
	dataType_->state_count = 4;
}

GeneralDataTypeInterface::GeneralDataTypeInterface(const std::vector<std::string> &states,
                                                   std::optional<const std::map<std::string, std::vector<std::string>>> ambiguities) {
	if (states.empty()) throw Error("general data type needs at least one state");
	dataType_->kind = phyamd::DataTypeKind::General;
	dataType_->state_count = (int)states.size();
	dataType_->symbol_length = (int)states[0].size();
	dataType_->states = states;
	// named ambiguity sets (datatype.c:243-262): exact whenever tip partials are used (4 states: tip masks; 20 / 60 / 61:
	// the engine's set codes); with tip states they read as "unknown", as every code >= the state count does in the reference
	if (ambiguities.has_value())
		for (const auto &kv : *ambiguities) {
			std::vector<int> set;
			for (const auto &name : kv.second) {
				auto it = std::find(states.begin(), states.end(), name);
				if (it == states.end()) throw Error("ambiguity `" + kv.first + "` refers to unknown state `" + name + "`");
				set.push_back((int)(it - states.begin()));
			}
			dataType_->ambiguities.emplace_back(kv.first, set);
		}
}

// ---------------------------------------------------------------------------------------------
// tree models
// ---------------------------------------------------------------------------------------------
TreeModelInterface::~TreeModelInterface() = default;

void TreeModelInterface::InitializeMap(const std::vector<std::string> &taxa) {
	// physher.cpp:15-32: tips map to their index in the taxon list, internal nodes to class id + tip count
	nodeMap_.resize(nodeCount_);
	for (size_t i = 0; i < nodeCount_; i++) {
		if (tree_->is_leaf((int)i)) {
			auto it = std::find(taxa.begin(), taxa.end(), tree_->name[i]);
			nodeMap_[i] = (size_t)(it - taxa.begin());
		} else
			nodeMap_[i] = (size_t)tree_->class_id[i] + tipCount_;
	}
}

UnRootedTreeModelInterface::UnRootedTreeModelInterface(const std::string &newick, const std::vector<std::string> &taxa) {
	tree_ = std::make_unique<phyamd::Tree>(phyamd::make_unrooted_tree(newick, taxa));
	nodeCount_ = (size_t)tree_->node_count;
	tipCount_ = (size_t)tree_->tip_count;
	InitializeMap(taxa);
	parameterCount_ = nodeCount_ - 2;
}

void UnRootedTreeModelInterface::SetParameters(const double *parameters) {
	auto &t = *tree_;
	const int root = t.root, rl = t.left[root], rr = t.right[root];
	for (int n = 0; n < t.node_count; n++) {
		if (n == root || n == rl || n == rr) continue;
		t.distance[n] = parameters[nodeMap_[n]];
	}
	// the merged root branch lives on root->left (physher.cpp:59-63)
	t.distance[rl] = parameters[nodeMap_[t.is_leaf(rr) ? rr : rl]];
	version_++;
}

void UnRootedTreeModelInterface::GetParameters(double *parameters) {
	auto &t = *tree_;
	const int root = t.root, rl = t.left[root], rr = t.right[root];
	for (int n = 0; n < t.node_count; n++) {
		if (n == root || n == rl || n == rr) continue;
		parameters[nodeMap_[n]] = t.distance[n];
	}
	parameters[nodeMap_[t.is_leaf(rr) ? rr : rl]] = t.distance[rl];
}

TimeTreeModelInterface::TimeTreeModelInterface(const std::string &newick, const std::vector<std::string> &taxa, const std::vector<double> dates) {
	tree_ = std::make_unique<phyamd::Tree>(phyamd::make_time_tree(newick, taxa, dates));
	nodeCount_ = (size_t)tree_->node_count;
	tipCount_ = (size_t)tree_->tip_count;
	InitializeMap(taxa);
	parameterCount_ = tipCount_ - 1;
}

void TimeTreeModelInterface::SetParameters(const double *parameters) {
	auto &t = *tree_;
	for (int n = t.tip_count; n < t.node_count; n++) t.height[n] = parameters[t.class_id[n]];
	version_++;
}

void TimeTreeModelInterface::GetParameters(double *parameters) {
	auto &t = *tree_;
	for (int n = t.tip_count; n < t.node_count; n++) parameters[t.class_id[n]] = t.height[n];
}

void TimeTreeModelInterface::GetNodeHeights(double *heights) { TimeTreeModelInterface::GetParameters(heights); }

ReparameterizedTimeTreeModelInterface::ReparameterizedTimeTreeModelInterface(const std::string &newick, const std::vector<std::string> &taxa,
                                                                             const std::vector<double> dates, TreeTransformFlags transform)
    : TimeTreeModelInterface(newick, taxa, dates) {
	if (transform != TreeTransformFlags::RATIO) throw Error("only the RATIO node-height transform is built (treetransform.h:19)");
	phyamd::enable_ratio_transform(*tree_);
}

void ReparameterizedTimeTreeModelInterface::SetParameters(const double *parameters) {
	auto &t = *tree_;
	std::copy(parameters, parameters + t.tip_count - 1, t.ratios.begin());
	phyamd::heights_from_ratios(t);
	version_++;
}

void ReparameterizedTimeTreeModelInterface::GetParameters(double *parameters) {
	std::copy(tree_->ratios.begin(), tree_->ratios.end(), parameters);
}

void ReparameterizedTimeTreeModelInterface::GetNodeHeights(double *heights) { TimeTreeModelInterface::GetParameters(heights); }

void ReparameterizedTimeTreeModelInterface::GradientTransformJVP(double *gradient, const double *height_gradient) {
	phyamd::ratio_transform_jvp(*tree_, height_gradient, gradient);
}

void ReparameterizedTimeTreeModelInterface::GradientTransformJVP(double *gradient, const double *height_gradient, const double *heights) {
	phyamd::Tree tmp = *tree_;  // Tree_node_transform_jvp_with_heights: evaluate at the caller's heights
	for (int n = tmp.tip_count; n < tmp.node_count; n++) tmp.height[n] = heights[tmp.class_id[n]];
	for (int n = tmp.tip_count; n < tmp.node_count; n++)
		tmp.ratios[tmp.class_id[n]] = n == tmp.root ? tmp.height[n] : (tmp.height[n] - tmp.lowers[n]) / (tmp.height[tmp.parent[n]] - tmp.lowers[n]);
	phyamd::ratio_transform_jvp(tmp, height_gradient, gradient);
}

void ReparameterizedTimeTreeModelInterface::GradientTransformJacobian(double *gradient) {
	std::fill(gradient, gradient + tipCount_ - 1, 0.0);
	phyamd::ratio_transform_log_jacobian_gradient(*tree_, gradient);
}

double ReparameterizedTimeTreeModelInterface::TransformJacobian() { return phyamd::ratio_transform_log_jacobian(*tree_); }

// ---------------------------------------------------------------------------------------------
// clock models
// ---------------------------------------------------------------------------------------------
void BranchModelInterface::SetParameters(const double *parameters) { SetRates(parameters); }
void BranchModelInterface::GetParameters(double *parameters) { std::copy(rates_.begin(), rates_.end(), parameters); }
void BranchModelInterface::SetRates(const double *rates) {
	std::copy(rates, rates + rates_.size(), rates_.begin());
	version_++;
}
double BranchModelInterface::Rate(size_t node_id) const { return rates_[map_[node_id]]; }

StrictClockModelInterface::StrictClockModelInterface(double rate, TreeModelInterface *treeModel) {
	treeModel_ = treeModel;
	rates_ = {rate};
	map_.assign(treeModel->GetNodeCount(), 0);
	parameterCount_ = 1;
}
void StrictClockModelInterface::SetRate(double rate) {
	rates_[0] = rate;
	version_++;
}

SimpleClockModelInterface::SimpleClockModelInterface(const std::vector<double> &rates, TreeModelInterface *treeModel) {
	treeModel_ = treeModel;
	const size_t N = treeModel->GetNodeCount();
	if (rates.size() != N - 1) throw Error("SimpleClockModelInterface needs nodeCount - 1 rates");
	rates_ = rates;
	map_.assign(N, 0);
	const phyamd::Tree &t = *treeModel->GetTree();
	for (size_t n = 0; n < N; n++)  // physher.cpp:206-213: rate index = taxon index for tips, class id + tip count for clades
		if ((int)n != t.root) map_[n] = t.is_leaf((int)n) ? (size_t)t.class_id[n] : (size_t)t.class_id[n] + treeModel->GetTipCount();
	parameterCount_ = N - 1;
}

// ---------------------------------------------------------------------------------------------
// substitution models
// ---------------------------------------------------------------------------------------------
SubstitutionModelInterface::~SubstitutionModelInterface() {
	if (ownsDataType_) delete dataType_;
}

JC69Interface::JC69Interface() {
	dataType_ = new NucleotideDataTypeInterface();
	ownsDataType_ = true;
	substModel_ = std::make_unique<phyamd::SubstModel>();
	substModel_->name = "JC69";
	substModel_->freqs = {0.25, 0.25, 0.25, 0.25};
	parameterCount_ = 0;
}

HKYInterface::HKYInterface(double kappa, const std::vector<double> &frequencies) {
	dataType_ = new NucleotideDataTypeInterface();
	ownsDataType_ = true;
	substModel_ = std::make_unique<phyamd::SubstModel>();
	substModel_->name = "HKY";
	substModel_->rates = {kappa};
	substModel_->freqs = frequencies;
	parameterCount_ = 5;
}
void HKYInterface::SetKappa(double kappa) {
	substModel_->rates[0] = kappa;
	substModel_->dirty = true;
	version_++;
}
void HKYInterface::SetFrequencies(const double *f) {
	std::copy(f, f + 4, substModel_->freqs.begin());
	substModel_->dirty = true;
	version_++;
}
void HKYInterface::SetParameters(const double *parameters) {
	// the reference reads kappa from parameters[1] and the frequencies from parameters + 1 (physher.cpp:275-278); kept
	SetKappa(parameters[1]);
	SetFrequencies(parameters + 1);
}

GTRInterface::GTRInterface(const std::vector<double> &rates, const std::vector<double> &frequencies) {
	if (rates.size() != 5 && rates.size() != 6) throw Error("GTRInterface takes 5 rates (relative to GT) or a 6-rate simplex");
	dataType_ = new NucleotideDataTypeInterface();
	ownsDataType_ = true;
	substModel_ = std::make_unique<phyamd::SubstModel>();
	substModel_->name = "GTR";
	substModel_->rates = rates;
	substModel_->freqs = frequencies;
	parameterCount_ = rates.size() == 6 ? 10 : 9;
}
void GTRInterface::SetRates(const double *rates) {
	std::copy(rates, rates + substModel_->rates.size(), substModel_->rates.begin());
	substModel_->dirty = true;
	version_++;
}
void GTRInterface::SetFrequencies(const double *f) {
	std::copy(f, f + 4, substModel_->freqs.begin());
	substModel_->dirty = true;
	version_++;
}
void GTRInterface::SetParameters(const double *parameters) {
	// physher.cpp:316-319: rates first, frequencies from parameters + 3 (sic)
	SetRates(parameters);
	SetFrequencies(parameters + 3);
}

GeneralSubstitutionModelInterface::GeneralSubstitutionModelInterface(DataTypeInterface *dataType, const std::vector<double> &rates,
                                                                     const std::vector<double> &frequencies, const std::vector<unsigned> &mapping,
                                                                     bool normalize) {
	dataType_ = dataType;
	substModel_ = std::make_unique<phyamd::SubstModel>();
	substModel_->name = "GENERAL";
	substModel_->S = (int)frequencies.size();
	substModel_->rates = rates;
	substModel_->freqs = frequencies;
	substModel_->structure = mapping;
	substModel_->normalize = normalize;
	parameterCount_ = rates.size();
}
void GeneralSubstitutionModelInterface::SetRates(const double *rates) {
	std::copy(rates, rates + substModel_->rates.size(), substModel_->rates.begin());
	substModel_->dirty = true;
	version_++;
}
void GeneralSubstitutionModelInterface::SetFrequencies(const double *f) {
	std::copy(f, f + substModel_->S, substModel_->freqs.begin());
	substModel_->dirty = true;
	version_++;
}
void GeneralSubstitutionModelInterface::SetParameters(const double *parameters) {
	SetRates(parameters);
	SetFrequencies(parameters + 3);  // physher.cpp:364-367 (sic)
}

// ---------------------------------------------------------------------------------------------
// site models
// ---------------------------------------------------------------------------------------------
SiteModelInterface::~SiteModelInterface() = default;

void SiteModelInterface::SetMu(double mu) {
	siteModel_->mu = mu;
	version_++;
}
void SiteModelInterface::GetRates(double *rates) {
	siteModel_->update();
	std::copy(siteModel_->cat_rates.begin(), siteModel_->cat_rates.end(), rates);
}
void SiteModelInterface::GetProportions(double *p) {
	siteModel_->update();
	std::copy(siteModel_->cat_props.begin(), siteModel_->cat_props.end(), p);
}
void SiteModelInterface::SetParameters(const double *parameters) {
	size_t k = 0;
	if (siteModel_->dist != phyamd::RateDistribution::Constant) siteModel_->shape = parameters[k++];
	if (siteModel_->has_pinv) siteModel_->pinv = parameters[k++];
	if (siteModel_->has_mu) siteModel_->mu = parameters[k++];
	siteModel_->dirty = true;
	version_++;
}
void SiteModelInterface::GetParameters(double *parameters) {
	size_t k = 0;
	if (siteModel_->dist != phyamd::RateDistribution::Constant) parameters[k++] = siteModel_->shape;
	if (siteModel_->has_pinv) parameters[k++] = siteModel_->pinv;
	if (siteModel_->has_mu) parameters[k++] = siteModel_->mu;
}

ConstantSiteModelInterface::ConstantSiteModelInterface(std::optional<double> mu) {
	siteModel_ = std::make_unique<phyamd::SiteModel>();
	if (mu.has_value()) {
		siteModel_->has_mu = true;
		siteModel_->mu = *mu;
	}
	parameterCount_ = mu.has_value() ? 1 : 0;
}
void ConstantSiteModelInterface::GetRates(double *rates) { rates[0] = siteModel_->has_mu ? siteModel_->mu : 1.0; }

InvariantSiteModelInterface::InvariantSiteModelInterface(double proportionInvariant, std::optional<double> mu) {
	siteModel_ = std::make_unique<phyamd::SiteModel>();
	siteModel_->cat_count = 2;
	siteModel_->has_pinv = true;
	siteModel_->pinv = proportionInvariant;
	if (mu.has_value()) {
		siteModel_->has_mu = true;
		siteModel_->mu = *mu;
	}
	parameterCount_ = 1 + (mu.has_value() ? 1 : 0);
}
void InvariantSiteModelInterface::SetProportionInvariant(double v) {
	siteModel_->pinv = v;
	siteModel_->dirty = true;
	version_++;
}

DiscretizedSiteModelInterface::DiscretizedSiteModelInterface(int distribution, double shape, size_t categories,
                                                             std::optional<double> proportionInvariant, std::optional<double> mu) {
	siteModel_ = std::make_unique<phyamd::SiteModel>();
	siteModel_->dist = distribution == 0 ? phyamd::RateDistribution::Gamma : phyamd::RateDistribution::Weibull;
	siteModel_->shape = shape;
	// new_SiteModel_with_parameters adds the invariant class on top of `categories` (sitemodel.c constructor)
	siteModel_->cat_count = (int)categories + (proportionInvariant.has_value() ? 1 : 0);
	if (proportionInvariant.has_value()) {
		siteModel_->has_pinv = true;
		siteModel_->pinv = *proportionInvariant;
	}
	if (mu.has_value()) {
		siteModel_->has_mu = true;
		siteModel_->mu = *mu;
	}
	parameterCount_ = 1 + (proportionInvariant.has_value() ? 1 : 0) + (mu.has_value() ? 1 : 0);
	categoryCount_ = (size_t)siteModel_->cat_count;
}
void DiscretizedSiteModelInterface::SetParameter(double p) {
	siteModel_->shape = p;
	siteModel_->dirty = true;
	version_++;
}
void DiscretizedSiteModelInterface::SetProportionInvariant(double v) {
	siteModel_->pinv = v;
	siteModel_->dirty = true;
	version_++;
}

WeibullSiteModelInterface::WeibullSiteModelInterface(double shape, size_t categories, std::optional<double> pinv, std::optional<double> mu)
    : DiscretizedSiteModelInterface(1, shape, categories, pinv, mu) {}
void WeibullSiteModelInterface::SetShape(double shape) { SetParameter(shape); }

GammaSiteModelInterface::GammaSiteModelInterface(double shape, size_t categories, std::optional<double> pinv, std::optional<double> mu)
    : DiscretizedSiteModelInterface(0, shape, categories, pinv, mu) {}
void GammaSiteModelInterface::SetShape(double shape) { SetParameter(shape); }
void GammaSiteModelInterface::SetEpsilon(double epsilon) { epsilon_ = epsilon; }

// ---------------------------------------------------------------------------------------------
// tree likelihood
// ---------------------------------------------------------------------------------------------
namespace phyamd {

struct LikelihoodImpl {
	phyamd_engine *engine = nullptr;
	Patterns patterns;
	unsigned long tree_v = ~0ul, subst_v = ~0ul, site_v = ~0ul, clock_v = ~0ul;
	int S = 0, Sp = 0;  // model states and the engine's (padded) state count
	unsigned long dq_v = ~0ul;  // substitution-model version the uploaded dQ/dtheta belong to
	bool dq_rates = false, dq_freqs = false;
	std::vector<double> branch_lengths, cat_grad;
	std::vector<double> sent_lengths;  // what the engine holds
	~LikelihoodImpl() {
		if (engine) phyamd_destroy(engine);
	}
};

static void check(int rc) {
	if (rc != PHYAMD_OK) throw Error(std::string("physher_amd engine: ") + phyamd_last_error());
}

// (lives here, not in patterns.cpp: that file stays free of the engine so the host sanitiser build needs no device library)
Patterns compress_patterns_device(const DataType &dt, const std::vector<std::string> &names, const std::vector<std::string> &sequences) {
	const int T = (int)sequences.size();
	if (T == 0 || names.size() != sequences.size()) throw Error("alignment needs one name per sequence");
	const size_t len = sequences[0].size();
	for (const auto &s : sequences)
		if (s.size() != len) throw Error("sequences are not aligned (different lengths)");
	const int step = dt.symbol_length;
	const size_t sites = len / step;
	if (sites == 0) throw Error("empty alignment");
	std::vector<const uint8_t *> rows(T);
	std::vector<uint8_t> lut, coded;
	if (step == 1) {  // one-byte symbols: the device translates through the data type's table
		lut.resize(256);
		char sym[2] = {0, 0};
		for (int ch = 0; ch < 256; ch++) {
			sym[0] = (char)ch;
			lut[ch] = (uint8_t)dt.encode(sym);
		}
		for (int t = 0; t < T; t++) rows[t] = reinterpret_cast<const uint8_t *>(sequences[t].data());
	} else {  // codons and other multi-character symbols are coded here
		coded.resize((size_t)T * sites);
		for (int t = 0; t < T; t++) {
			for (size_t s = 0; s < sites; s++) coded[(size_t)t * sites + s] = (uint8_t)dt.encode(sequences[t].data() + s * step);
			rows[t] = &coded[(size_t)t * sites];
		}
	}
	Patterns p;
	p.taxon_count = T;
	p.site_count = (int)sites;
	p.names = names;
	p.states.resize((size_t)T * sites);
	p.weights.resize(sites);
	int32_t count = 0;
	if (phyamd_compress_patterns(-1, T, (int64_t)sites, rows.data(), lut.empty() ? nullptr : lut.data(), &count, p.states.data(), p.weights.data()) != PHYAMD_OK)
		throw Error(phyamd_last_error());
	p.pattern_count = count;
	p.states.resize((size_t)T * count);
	p.weights.resize(count);
	return p;
}

}  // namespace phyamd

// state counts the engine has kernels for; other counts are padded up with states nothing can enter or leave
static int padded_state_count(int S) {
	if (S <= 4) return 4;
	if (S <= 20) return 20;
	if (S <= 60) return 60;
	if (S == 61) return 61;
	throw Error("more than 61 states are not supported by the device engine");
}

TreeLikelihoodInterface::TreeLikelihoodInterface(const std::vector<std::pair<std::string, std::string>> &alignment, TreeModelInterface *treeModel,
                                                 SubstitutionModelInterface *substitutionModel, SiteModelInterface *siteModel,
                                                 std::optional<BranchModelInterface *> branchModel, bool use_ambiguities, bool use_tip_states,
                                                 bool include_jacobian)
    : treeModel_(treeModel),
      substitutionModel_(substitutionModel),
      siteModel_(siteModel),
      branchModel_(branchModel.has_value() ? *branchModel : nullptr),
      includeJacobian_(include_jacobian),
      impl_(std::make_unique<phyamd::LikelihoodImpl>()) {
	(void)use_ambiguities;
	const phyamd::DataType &dt = *substitutionModel->GetDataType()->dataType_;
	std::vector<std::string> names, seqs;
	for (const auto &kv : alignment) {
		names.push_back(kv.first);
		seqs.push_back(kv.second);
	}
	// new_SitePattern (physher.cpp:569-577); long alignments on the device (same patterns in the same order)
	bool compressed = false;
	if (!seqs.empty() && seqs[0].size() / (size_t)dt.symbol_length >= phyamd::kDeviceCompressionSites) {
		try {
			impl_->patterns = phyamd::compress_patterns_device(dt, names, seqs);
			compressed = true;
		} catch (const phyamd::Error &) {  // two columns collided in both hashes: the sequential table is exact by construction
		}
	}
	if (!compressed) impl_->patterns = phyamd::compress_patterns(dt, names, seqs);
	Init(use_tip_states);
}

// new_AttributePattern (sitepattern.c:321-353, physher.cpp:594-629): ONE pattern of weight 1, one attribute (a state name of
// the data type) per taxon -- discrete-trait / phylogeography likelihoods
TreeLikelihoodInterface::TreeLikelihoodInterface(const std::vector<std::string> &taxa, const std::vector<std::string> &attributes,
                                                 TreeModelInterface *treeModel, SubstitutionModelInterface *substitutionModel,
                                                 SiteModelInterface *siteModel, std::optional<BranchModelInterface *> branchModel, bool use_ambiguities,
                                                 bool use_tip_states, bool include_jacobian)
    : treeModel_(treeModel),
      substitutionModel_(substitutionModel),
      siteModel_(siteModel),
      branchModel_(branchModel.has_value() ? *branchModel : nullptr),
      includeJacobian_(include_jacobian),
      impl_(std::make_unique<phyamd::LikelihoodImpl>()) {
	(void)use_ambiguities;
	if (taxa.size() != attributes.size()) throw Error("taxa and attributes differ in length");
	const phyamd::DataType &dt = *substitutionModel->GetDataType()->dataType_;
	phyamd::Patterns &pt = impl_->patterns;
	pt.taxon_count = (int)taxa.size();
	pt.pattern_count = pt.site_count = 1;
	pt.names = taxa;
	pt.weights.assign(1, 1.0);
	pt.states.resize(taxa.size());
	for (size_t i = 0; i < taxa.size(); i++) {
		const int code = dt.encode_string(attributes[i]);
		if (code > 255) throw Error("state code does not fit the pattern array");
		pt.states[i] = (uint8_t)code;
	}
	Init(use_tip_states);
}

void TreeLikelihoodInterface::Init(bool use_tip_states) {
	const phyamd::DataType &dt = *substitutionModel_->GetDataType()->dataType_;
	const std::vector<std::string> &names = impl_->patterns.names;
	const phyamd::Tree &t = *treeModel_->GetTree();
	if ((int)names.size() != t.tip_count) throw Error("alignment and tree have different numbers of taxa");
	const int S = dt.state_count, P = impl_->patterns.pattern_count;
	if (substitutionModel_->GetModel()->S != S) throw Error("substitution model and data type differ in state count");
	const int Sp = padded_state_count(S);
	impl_->S = S;
	impl_->Sp = Sp;
	siteModel_->GetModel()->update();
	phyamd_config cfg{};
	cfg.tip_count = t.tip_count;
	cfg.pattern_count = P;
	cfg.state_count = Sp;
	cfg.category_count = siteModel_->GetModel()->cat_count;
	cfg.device = -1;
	cfg.rescale = PHYAMD_RESCALE_AUTO;
	phyamd::check(phyamd_create(&cfg, &impl_->engine));
	phyamd::check(phyamd_set_topology(impl_->engine, t.left.data(), t.right.data(), t.root));
	phyamd::check(phyamd_set_pattern_weights(impl_->engine, impl_->patterns.weights.data()));
	// tlk->mapping: node -> sequence by NAME (treelikelihood.c:1095-1104)
	std::vector<double> partial((size_t)P * Sp, 0.0), one(S);
	std::vector<uint8_t> codes_p(P);
	for (int tip = 0; tip < t.tip_count; tip++) {
		auto it = std::find(names.begin(), names.end(), t.name[tip]);
		if (it == names.end()) throw Error("Could not find taxon `" + t.name[tip] + "` in alignment");
		const uint8_t *codes = &impl_->patterns.states[(size_t)(it - names.begin()) * P];
		if (!use_tip_states) {
			// "tipstates": false -- datatype->partial per pattern (treelikelihood.c:1106-1117): ambiguity codes and named sets are
			// exact; states added by padding stay 0 (on the 4-state engine the 0/1 vector becomes the tip's 4-bit mask)
			for (int k = 0; k < P; k++) {
				dt.partial(codes[k], one.data());
				for (int i = 0; i < Sp; i++) partial[(size_t)k * Sp + i] = i < S ? one[i] : 0.0;
			}
			phyamd::check(phyamd_set_tip_partials(impl_->engine, tip, partial.data()));
		} else {
			// tip states: every code >= S (unknown, ambiguity codes, named sets) reads as "all states" in the reference's
			// kernels too (sitepattern.h:68-82)
			for (int k = 0; k < P; k++) codes_p[k] = codes[k] >= S ? (uint8_t)Sp : codes[k];
			phyamd::check(phyamd_set_tip_states(impl_->engine, tip, codes_p.data()));
		}
	}
	parameterCount_ = 0;
	RequestGradient();
}

TreeLikelihoodInterface::~TreeLikelihoodInterface() = default;

size_t TreeLikelihoodInterface::GetPatternCount() const { return (size_t)impl_->patterns.pattern_count; }
const std::vector<double> &TreeLikelihoodInterface::PatternWeights() const { return impl_->patterns.weights; }
const std::vector<unsigned char> &TreeLikelihoodInterface::PatternStates() const { return impl_->patterns.states; }

void TreeLikelihoodInterface::RequestGradient(std::vector<TreeLikelihoodGradientFlags> flags) {
	int f = 0;
	for (auto x : flags) f |= (int)x;
	const phyamd::SiteModel &smc = *siteModel_->GetModel();
	const bool site_params = smc.dist != phyamd::RateDistribution::Constant || smc.has_pinv || smc.has_mu;
	const phyamd::SubstModel &mc = *substitutionModel_->GetModel();
	const bool subst_params = mc.name != "JC69";  // m->dPdp != NULL (gtr.c:102, hky.c:72, gensubst.c:189)
	if (f == 0) {  // TreeLikelihood_initialize_gradient(flags = 0): everything differentiable (treelikelihood.c:255-270)
		f = (int)TreeLikelihoodGradientFlags::TREE_HEIGHT;
		if (site_params) f |= (int)TreeLikelihoodGradientFlags::SITE_MODEL;
		if (branchModel_) f |= (int)TreeLikelihoodGradientFlags::BRANCH_MODEL;
		if (subst_params) f |= (int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL;
	}
	const int subst_flags = (int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL | (int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL_RATES |
	                        (int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL_FREQUENCIES;
	if ((f & subst_flags) && !subst_params)
		throw Error("this substitution model has no differentiable parameters (no dPdp in the reference either)");
	flags_ = f;
	// treelikelihood.c:247-249: SUBSTITUTION_MODEL = rates + frequencies
	substRates_ = f & ((int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL | (int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL_RATES);
	substFreqs_ = f & ((int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL | (int)TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL_FREQUENCIES);
	const phyamd::Tree &t = *treeModel_->GetTree();
	size_t len = 0;
	if (f & (int)TreeLikelihoodGradientFlags::TREE_HEIGHT) len += t.time_mode ? (size_t)t.tip_count - 1 : (size_t)t.node_count;
	if (f & (int)TreeLikelihoodGradientFlags::SITE_MODEL)  // treelikelihood.c:279-287: shape, pinv, mu
		len += (smc.dist != phyamd::RateDistribution::Constant) + (smc.has_pinv ? 1 : 0) + (smc.has_mu ? 1 : 0);
	if ((f & (int)TreeLikelihoodGradientFlags::BRANCH_MODEL) && branchModel_) len += branchModel_->rates_.size();
	if (substRates_) len += (size_t)mc.rate_parameter_count();  // treelikelihood.c:296-305: constrained values, K of a simplex
	if (substFreqs_) len += (size_t)mc.S;
	gradientLength_ = len;
	// physher.cpp:639-641 drops the two root-adjacent entries of the unrooted tree block; the reference subtracts them even
	// when no tree block was requested (and then mis-copies in Gradient, physher.cpp:648-655) -- here only with a tree block
	if (branchModel_ == nullptr && (f & (int)TreeLikelihoodGradientFlags::TREE_HEIGHT)) gradientLength_ -= 2;
}

void TreeLikelihoodInterface::Sync() {
	auto &I = *impl_;
	phyamd::Tree &t = *treeModel_->GetTree();
	const unsigned long cv = branchModel_ ? branchModel_->version_ : 0;
	if (I.tree_v != treeModel_->version_ || I.clock_v != cv) {
		I.branch_lengths.assign(t.node_count, 0.0);
		for (int n = 0; n < t.node_count; n++) {
			if (n == t.root) continue;
			if (t.time_mode) {
				const double rate = branchModel_ ? branchModel_->Rate((size_t)n) : 1.0;
				const double bl = rate * (t.height[t.parent[n]] - t.height[n]);  // treelikelihood.c:1657
				if (bl < 0) throw Error("negative branch length above node " + std::to_string(n));
				I.branch_lengths[n] = bl;
			} else
				I.branch_lengths[n] = t.distance[n];
		}
		// Parameters_set_values only fires listeners for values that changed (update_nodes[index], treelikelihood.c:73-92):
		// a few changed branches are sent one by one and the engine recomputes only the paths above them
		size_t diffs = 0;
		const bool have_prev = I.sent_lengths.size() == I.branch_lengths.size();
		if (have_prev)
			for (int n = 0; n < t.node_count; n++) diffs += I.sent_lengths[n] != I.branch_lengths[n];
		if (have_prev && diffs > 0 && diffs <= (size_t)t.node_count / 4) {
			for (int n = 0; n < t.node_count; n++)
				if (I.sent_lengths[n] != I.branch_lengths[n]) phyamd::check(phyamd_set_branch_length(I.engine, n, I.branch_lengths[n]));
		} else if (!have_prev || diffs > 0)
			phyamd::check(phyamd_set_branch_lengths(I.engine, I.branch_lengths.data()));
		I.sent_lengths = I.branch_lengths;
		I.tree_v = treeModel_->version_;
		I.clock_v = cv;
	}
	if (I.subst_v != substitutionModel_->version_) {
		phyamd::SubstModel &m = *substitutionModel_->GetModel();
		m.update();
		if (I.Sp == I.S) {
			phyamd::check(phyamd_set_eigen(I.engine, m.eval.data(), m.evec.data(), m.ivec.data()));
			phyamd::check(phyamd_set_frequencies(I.engine, m.freqs.data()));
		} else {
			// padding states: eigenvalue 0 with unit eigenvectors (P(t) = identity on them), frequency 0, tip partial 0 -- nothing
			// enters or leaves them and they carry no weight at the root, so lnL and every gradient are those of the S-state model
			const int S = I.S, Sp = I.Sp;
			std::vector<double> ev(Sp, 0.0), U((size_t)Sp * Sp, 0.0), Ui((size_t)Sp * Sp, 0.0), f(Sp, 0.0);
			for (int i = 0; i < Sp; i++) U[(size_t)i * Sp + i] = Ui[(size_t)i * Sp + i] = 1.0;
			for (int i = 0; i < S; i++) {
				ev[i] = m.eval[i];
				f[i] = m.freqs[i];
				for (int j = 0; j < S; j++) {
					U[(size_t)i * Sp + j] = m.evec[(size_t)i * S + j];
					Ui[(size_t)i * Sp + j] = m.ivec[(size_t)i * S + j];
				}
			}
			phyamd::check(phyamd_set_eigen(I.engine, ev.data(), U.data(), Ui.data()));
			phyamd::check(phyamd_set_frequencies(I.engine, f.data()));
		}
		I.subst_v = substitutionModel_->version_;
	}
	if (I.site_v != siteModel_->version_) {
		phyamd::SiteModel &sm = *siteModel_->GetModel();
		sm.update();
		std::vector<double> r(sm.cat_count);
		for (int c = 0; c < sm.cat_count; c++) r[c] = sm.rate(c);  // get_rate includes mu (sitemodel.c:544-549)
		phyamd::check(phyamd_set_category_rates(I.engine, r.data(), sm.cat_props.data()));
		I.site_v = siteModel_->version_;
	}
}

double TreeLikelihoodInterface::LogLikelihood() {
	Sync();
	double lnl = 0.0;
	phyamd::check(phyamd_log_likelihood(impl_->engine, &lnl));
	const phyamd::Tree &t = *treeModel_->GetTree();
	if (includeJacobian_ && t.reparameterized) lnl += phyamd::ratio_transform_log_jacobian(t);  // treelikelihood.c:166-170
	return lnl;
}

void TreeLikelihoodInterface::Gradient(double *gradient) {
	Sync();
	auto &I = *impl_;
	const phyamd::Tree &t = *treeModel_->GetTree();
	phyamd::SiteModel &sm = *siteModel_->GetModel();
	const int N = t.node_count, C = sm.cat_count;
	I.cat_grad.assign((size_t)N * C, 0.0);
	double lnl = 0.0;
	const bool subst = substRates_ || substFreqs_;
	// a substitution-model request clears include_root_freqs in the reference too (treelikelihood.c:291-305)
	const int eflags = referenceCompat_ ? ((subst ? 0 : PHYAMD_GRAD_FOLD_ROOT_FREQS) | PHYAMD_GRAD_COMPAT_SCALED) : 0;
	std::vector<double> subst_grad;
	if (subst) {  // gradient_PMatrix (treelikelihood.c:3077-3110): all parameters in the same two passes as the branch gradient
		phyamd::SubstModel &m = *substitutionModel_->GetModel();
		if (I.dq_v != substitutionModel_->version_ || I.dq_rates != substRates_ || I.dq_freqs != substFreqs_) {
			std::vector<double> dQ;
			m.rate_matrix_derivatives(substRates_, substFreqs_, dQ);
			const int S = I.S, Sp = I.Sp, count = (int)(dQ.size() / ((size_t)S * S));
			if (Sp != S) {  // padding states: rows and columns of zeros
				std::vector<double> padded((size_t)count * Sp * Sp, 0.0);
				for (int th = 0; th < count; th++)
					for (int i = 0; i < S; i++)
						for (int j = 0; j < S; j++) padded[((size_t)th * Sp + i) * Sp + j] = dQ[((size_t)th * S + i) * S + j];
				dQ.swap(padded);
			}
			phyamd::check(phyamd_set_rate_matrix_derivatives(I.engine, count, dQ.data()));
			I.dq_v = substitutionModel_->version_;
			I.dq_rates = substRates_;
			I.dq_freqs = substFreqs_;
		}
		subst_grad.resize((substRates_ ? m.rate_parameter_count() : 0) + (substFreqs_ ? m.S : 0));
		phyamd::check(phyamd_parameter_gradient(I.engine, eflags, &lnl, I.cat_grad.data(), subst_grad.data()));
		if (substFreqs_) {  // + the root term d lnL / d pi_f (treelikelihood.c:2370-2401)
			std::vector<double> rootf(I.Sp);  // one entry per engine state; padding states come last
			phyamd::check(phyamd_root_frequency_term(I.engine, rootf.data()));
			for (int f = 0; f < m.S; f++) subst_grad[subst_grad.size() - m.S + f] += rootf[f];
		}
	} else
		phyamd::check(phyamd_gradient(I.engine, eflags, &lnl, I.cat_grad.data()));
	if (!t.time_mode)  // treelikelihood.c:3249-3255: the right child of the root carries no branch of an unrooted tree
		for (int c = 0; c < C; c++) I.cat_grad[(size_t)t.right[t.root] * C + c] = 0.0;
	// site-model parameters: gradient_discrete_sitemodel (treelikelihood.c:3010-3052) on the per-category gradients
	double site_grad[3];
	size_t site_count = 0;
	if (flags_ & (int)TreeLikelihoodGradientFlags::SITE_MODEL) {
		std::vector<double> ingrad(C, 0.0);
		const double mu = sm.has_mu ? sm.mu : 1.0;
		for (int n = 0; n < N; n++) {
			if (n == t.root || (!t.time_mode && n == t.right[t.root])) continue;
			for (int c = sm.has_pinv ? 1 : 0; c < C; c++) ingrad[c] += I.cat_grad[(size_t)n * C + c] * I.branch_lengths[n] * mu;  // :3237-3242
		}
		if (sm.dist != phyamd::RateDistribution::Constant) site_grad[site_count++] = sm.shape_gradient(ingrad.data());
		if (sm.has_pinv) {
			phyamd::check(phyamd_root_invariant_term(I.engine, &ingrad[0]));
			site_grad[site_count++] = sm.pinv_gradient(ingrad.data());
		}
	}
	// gradient_branch_length_from_cat_inplace (treelikelihood.c:3129-3143, 3258-3266).  The reference weights by
	// sm->cat_rates, i.e. WITHOUT mu, so its branch gradient is d lnL / d(mu * length); outside compatibility mode the
	// missing factor mu is applied and the result is d lnL / d length.
	std::vector<double> g(N, 0.0);
	const double mu_factor = (sm.has_mu && !referenceCompat_) ? sm.mu : 1.0;
	for (int n = 0; n < N; n++) {
		if (C == 1) g[n] = I.cat_grad[n] * mu_factor;
		else {
			double s = I.cat_grad[(size_t)n * C] * sm.cat_props[0] * sm.cat_rates[0];
			for (int c = 1; c < C; c++) s += I.cat_grad[(size_t)n * C + c] * sm.cat_props[c] * sm.cat_rates[c];
			g[n] = s * mu_factor;
		}
	}
	size_t j = 0;
	if (flags_ & (int)TreeLikelihoodGradientFlags::TREE_HEIGHT) {
		if (!t.time_mode) {
			const int rl = t.left[t.root], rr = t.right[t.root];
			// treelikelihood.c:3249-3255 zeroes root->right.  With a bifurcating-root newick whose right child is a tip, the
			// merged root branch is parameter nodeMap_[rr] (physher.cpp:59-61), so the reference reports 0 for a live
			// parameter; outside compatibility mode its derivative (that of root->left) is reported there instead.
			g[rr] = (t.is_leaf(rr) && !referenceCompat_) ? g[rl] : 0.0;
			for (int i = 0; i < N - 2; i++) gradient[treeModel_->nodeMap_[i]] = g[i];  // physher.cpp:649-655
			j = (size_t)N - 2;
		} else {
			// gradient_heights (treelikelihood.c:3145-3156): d bl_n = rate_n (d h_parent - d h_n)
			std::vector<double> gh(t.tip_count - 1, 0.0);
			for (int n : t.preorder) {
				if (n == t.root) continue;
				const double ng = g[n] * (branchModel_ ? branchModel_->Rate((size_t)n) : 1.0);
				if (!t.is_leaf(n)) gh[t.class_id[n]] -= ng;
				gh[t.class_id[t.parent[n]]] += ng;
			}
			if (t.reparameterized) {  // gradient_ratios (treelikelihood.c:3161-3171)
				phyamd::ratio_transform_jvp(t, gh.data(), gradient);
				if (includeJacobian_) phyamd::ratio_transform_log_jacobian_gradient(t, gradient);
			} else
				std::copy(gh.begin(), gh.end(), gradient);
			j = (size_t)t.tip_count - 1;
		}
	}
	if (flags_ & (int)TreeLikelihoodGradientFlags::SITE_MODEL) {
		for (size_t i = 0; i < site_count; i++) gradient[j++] = site_grad[i];
		if (sm.has_mu) {  // treelikelihood.c:3288-3302: sum over branches of d lnL / d(mu * length) times the length
			double gm = 0.0;
			for (int n = 0; n < N; n++)
				if (n != t.root) gm += g[n] / mu_factor * I.branch_lengths[n];
			gradient[j++] = gm;
		}
	}
	if ((flags_ & (int)TreeLikelihoodGradientFlags::BRANCH_MODEL) && branchModel_) {
		// gradient_clock (treelikelihood.c:3054-3075): d bl_n / d rate = elapsed time
		std::vector<double> gc(branchModel_->rates_.size(), 0.0);
		for (int n = 0; n < N; n++)
			if (n != t.root) gc[branchModel_->map_[n]] += g[n] * (t.height[t.parent[n]] - t.height[n]);
		for (double v : gc) gradient[j++] = v;
	}
	for (double v : subst_grad) gradient[j++] = v;  // [rates][frequencies], treelikelihood.c:3313-3357
	if (std::isnan(lnl) || std::isinf(lnl))
		for (size_t i = 0; i < j; i++) gradient[i] = NAN;  // treelikelihood.c:327-332
}
