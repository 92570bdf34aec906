// phycpp_amd/physher.hpp -- the C++ wrapper surface of physher's tree likelihood, MI355X-backed.
//
// Same class names, constructor arguments, method names and parameter/gradient orderings as the reference's
// src/phycpp/physher.hpp (the layer torchtree-physher binds), for the classes on the tree-likelihood path:
//   tree models      UnRootedTreeModelInterface, TimeTreeModelInterface, ReparameterizedTimeTreeModelInterface  (physher.hpp:127-174)
//   clock models     StrictClockModelInterface, SimpleClockModelInterface                                        (physher.hpp:176-199)
//   substitution     JC69Interface, HKYInterface, GTRInterface, GeneralSubstitutionModelInterface               (physher.hpp:201-267)
//   site models      Constant / Invariant / Weibull / Gamma SiteModelInterface                                  (physher.hpp:269-358)
//   likelihood       TreeLikelihoodInterface                                                                    (physher.hpp:360-395)
// The coalescent and CTMC-scale wrappers of the reference are priors, not part of this path, and are not provided.
// Behind TreeLikelihoodInterface sits the C ABI of include/physher_amd.h (HIP kernels); there is no CPU fallback.
#pragma once

#include <cstddef>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

namespace phyamd {
struct Tree;
struct SubstModel;
struct SiteModel;
struct DataType;
struct LikelihoodImpl;
}  // namespace phyamd

// bit values of the reference's TREELIKELIHOOD_FLAG_* (treelikelihood.h:32-38)
enum class TreeLikelihoodGradientFlags {
	TREE_HEIGHT = 1 << 0,
	SITE_MODEL = 1 << 1,
	SUBSTITUTION_MODEL = 1 << 2,
	SUBSTITUTION_MODEL_RATES = 1 << 3,
	SUBSTITUTION_MODEL_FREQUENCIES = 1 << 4,
	BRANCH_MODEL = 1 << 6
};

// treetransform.h:17-22
enum class TreeTransformFlags { RATIO = 1, SHIFT = 2, PROPORTION = 3 };

class DataTypeInterface {
   public:
	DataTypeInterface();
	virtual ~DataTypeInterface();
	std::shared_ptr<phyamd::DataType> dataType_;
};

class NucleotideDataTypeInterface : public DataTypeInterface {
   public:
	NucleotideDataTypeInterface();
};

class GeneralDataTypeInterface : public DataTypeInterface {
   public:
	GeneralDataTypeInterface(const std::vector<std::string> &states,
	                         std::optional<const std::map<std::string, std::vector<std::string>>> ambiguities);
};

class ModelInterface {
   public:
	virtual ~ModelInterface() = default;
	virtual void SetParameters(const double *parameters) = 0;
	virtual void GetParameters(double *parameters) = 0;
	size_t parameterCount_ = 0;
};

class CallableModelInterface : public ModelInterface {
   public:
	virtual double LogLikelihood() = 0;
	virtual void Gradient(double *gradient) = 0;
	size_t gradientLength_ = 0;
};

class TreeModelInterface : public ModelInterface {
   public:
	~TreeModelInterface() override;
	size_t GetNodeCount() { return nodeCount_; }
	size_t GetTipCount() { return tipCount_; }
	phyamd::Tree *GetTree() { return tree_.get(); }
	// version counter bumped on every change: the likelihood re-uploads branch lengths only when it moved
	unsigned long version_ = 0;
	std::vector<size_t> nodeMap_;

   protected:
	void InitializeMap(const std::vector<std::string> &taxa);
	std::unique_ptr<phyamd::Tree> tree_;
	size_t nodeCount_ = 0;
	size_t tipCount_ = 0;
};

class UnRootedTreeModelInterface : public TreeModelInterface {
   public:
	UnRootedTreeModelInterface(const std::string &newick, const std::vector<std::string> &taxa);
	// nodeCount - 2 branch lengths indexed by nodeMap_ (physher.cpp:51-64)
	void SetParameters(const double *parameters) override;
	void GetParameters(double *parameters) override;
};

class TimeTreeModelInterface : public TreeModelInterface {
   public:
	TimeTreeModelInterface(const std::string &newick, const std::vector<std::string> &taxa, const std::vector<double> dates);
	void SetParameters(const double *parameters) override;  // internal node heights by class id
	void GetParameters(double *parameters) override;
	virtual void GetNodeHeights(double *heights);
};

class ReparameterizedTimeTreeModelInterface : public TimeTreeModelInterface {
   public:
	ReparameterizedTimeTreeModelInterface(const std::string &newick, const std::vector<std::string> &taxa, const std::vector<double> dates,
	                                      TreeTransformFlags transform);
	void SetParameters(const double *parameters) override;  // ratios, root height at the root's class id
	void GetParameters(double *parameters) override;
	void GetNodeHeights(double *heights) override;
	void GradientTransformJVP(double *gradient, const double *height_gradient);
	void GradientTransformJVP(double *gradient, const double *height_gradient, const double *heights);
	void GradientTransformJacobian(double *gradient);
	double TransformJacobian();
};

class BranchModelInterface : public ModelInterface {
   public:
	void SetParameters(const double *parameters) override;
	void GetParameters(double *parameters) override;
	void SetRates(const double *rates);
	// rate of the branch above node `id`
	double Rate(size_t node_id) const;
	std::vector<double> rates_;
	std::vector<size_t> map_;  // node id -> index into rates_ (strict clock: all 0)
	unsigned long version_ = 0;

   protected:
	TreeModelInterface *treeModel_ = nullptr;
};

class StrictClockModelInterface : public BranchModelInterface {
   public:
	StrictClockModelInterface(double rate, TreeModelInterface *treeModel);
	void SetRate(double rate);
};

class SimpleClockModelInterface : public BranchModelInterface {
   public:
	SimpleClockModelInterface(const std::vector<double> &rates, TreeModelInterface *treeModel);
};

class SubstitutionModelInterface : public ModelInterface {
   public:
	~SubstitutionModelInterface() override;
	DataTypeInterface *GetDataType() { return dataType_; }
	phyamd::SubstModel *GetModel() { return substModel_.get(); }
	unsigned long version_ = 0;

   protected:
	std::unique_ptr<phyamd::SubstModel> substModel_;
	DataTypeInterface *dataType_ = nullptr;
	bool ownsDataType_ = false;
};

class JC69Interface : public SubstitutionModelInterface {
   public:
	JC69Interface();
	void SetParameters(const double *parameters) override {}
	void GetParameters(double *parameters) override {}
};

class HKYInterface : public SubstitutionModelInterface {
   public:
	HKYInterface(double kappa, const std::vector<double> &frequencies);
	void SetKappa(double kappa);
	void SetFrequencies(const double *frequencies);
	void SetParameters(const double *parameters) override;
	void GetParameters(double *parameters) override {}
};

class GTRInterface : public SubstitutionModelInterface {
   public:
	GTRInterface(const std::vector<double> &rates, const std::vector<double> &frequencies);
	void SetRates(const double *rates);
	void SetFrequencies(const double *frequencies);
	void SetParameters(const double *parameters) override;
	void GetParameters(double *parameters) override {}
};

class GeneralSubstitutionModelInterface : public SubstitutionModelInterface {
   public:
	GeneralSubstitutionModelInterface(DataTypeInterface *dataType, const std::vector<double> &rates, const std::vector<double> &frequencies,
	                                  const std::vector<unsigned> &mapping, bool normalize);
	void SetRates(const double *rates);
	void SetFrequencies(const double *frequencies);
	void SetParameters(const double *parameters) override;
	void GetParameters(double *parameters) override {}
};

class SiteModelInterface : public ModelInterface {
   public:
	~SiteModelInterface() override;
	void SetMu(double mu);
	virtual void GetRates(double *rates);
	virtual void GetProportions(double *proportions);
	void SetParameters(const double *parameters) override;  // [shape] [pinv] [mu], whichever exist (physher.cpp:504-519)
	void GetParameters(double *parameters) override;
	phyamd::SiteModel *GetModel() { return siteModel_.get(); }
	unsigned long version_ = 0;

   protected:
	std::unique_ptr<phyamd::SiteModel> siteModel_;
};

class ConstantSiteModelInterface : public SiteModelInterface {
   public:
	explicit ConstantSiteModelInterface(std::optional<double> mu);
	void GetRates(double *rates) override;  // includes mu (physher.cpp:393-395)
};

class InvariantSiteModelInterface : public SiteModelInterface {
   public:
	InvariantSiteModelInterface(double proportionInvariant, std::optional<double> mu);
	void SetProportionInvariant(double value);
};

class DiscretizedSiteModelInterface : public SiteModelInterface {
   public:
	void SetParameter(double parameter);
	void SetProportionInvariant(double value);
	size_t GetCategoryCount() { return categoryCount_; }

   protected:
	DiscretizedSiteModelInterface(int distribution, double shape, size_t categories, std::optional<double> proportionInvariant,
	                              std::optional<double> mu);
	size_t categoryCount_ = 1;
};

class WeibullSiteModelInterface : public DiscretizedSiteModelInterface {
   public:
	WeibullSiteModelInterface(double shape, size_t categories, std::optional<double> proportionInvariant, std::optional<double> mu);
	void SetShape(double shape);
};

class GammaSiteModelInterface : public DiscretizedSiteModelInterface {
   public:
	GammaSiteModelInterface(double shape, size_t categories, std::optional<double> proportionInvariant, std::optional<double> mu);
	void SetShape(double shape);
	void SetEpsilon(double epsilon);
	double epsilon_ = 1.e-6;
};

class TreeLikelihoodInterface : public CallableModelInterface {
   public:
	TreeLikelihoodInterface(const std::vector<std::pair<std::string, std::string>> &alignment, TreeModelInterface *treeModel,
	                        SubstitutionModelInterface *substitutionModel, SiteModelInterface *siteModel,
	                        std::optional<BranchModelInterface *> branchModel, bool use_ambiguities = false, bool use_tip_states = false,
	                        bool include_jacobian = false);
	// one attribute (state name) per taxon: discrete-trait likelihoods (physher.hpp:369-377, new_AttributePattern)
	TreeLikelihoodInterface(const std::vector<std::string> &taxa, const std::vector<std::string> &attributes, TreeModelInterface *treeModel,
	                        SubstitutionModelInterface *substitutionModel, SiteModelInterface *siteModel,
	                        std::optional<BranchModelInterface *> branchModel, bool use_ambiguities = false, bool use_tip_states = false,
	                        bool include_jacobian = false);
	~TreeLikelihoodInterface() override;

	void RequestGradient(std::vector<TreeLikelihoodGradientFlags> flags = std::vector<TreeLikelihoodGradientFlags>());
	double LogLikelihood() override;
	void Gradient(double *gradient) override;
	void SetParameters(const double *parameters) override {}
	void GetParameters(double *parameters) override {}
	void EnableSSE(bool flag) {}  // CPU kernel switch of the reference (physher.cpp:662-665): no meaning on the GPU path

	// --- beyond the reference surface ---
	// reproduce the reference's include_root_freqs = true / rescaled-gradient arithmetic bit for bit ("quirks")
	void SetReferenceCompatibility(bool on) { referenceCompat_ = on; }
	size_t GetPatternCount() const;
	const std::vector<double> &PatternWeights() const;
	const std::vector<unsigned char> &PatternStates() const;  // [taxon][pattern], taxa in alignment order

   private:
	void Init(bool use_tip_states);
	void Sync();
	TreeModelInterface *treeModel_;
	SubstitutionModelInterface *substitutionModel_;
	SiteModelInterface *siteModel_;
	BranchModelInterface *branchModel_;
	bool includeJacobian_;
	bool referenceCompat_ = false;
	int flags_ = 0;
	bool substRates_ = false, substFreqs_ = false;
	std::unique_ptr<phyamd::LikelihoodImpl> impl_;
};
