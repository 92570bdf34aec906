// pybind.cpp -- Python module over the phycpp-compatible classes (the role torchtree-physher's binding plays for
// the reference's src/phycpp).  Arrays cross as numpy float64; nothing is computed here.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "phyamd_host.hpp"
#include "phycpp_amd/physher.hpp"

namespace py = pybind11;
using darray = py::array_t<double, py::array::c_style | py::array::forcecast>;

static darray vec(const std::vector<double> &v) { return darray(v.size(), v.data()); }

PYBIND11_MODULE(_phycpp_amd, m) {
	m.doc() = "phycpp-compatible host classes over the MI355X tree-likelihood engine";
	py::register_exception<phyamd::Error>(m, "PhyamdError");

	py::enum_<TreeLikelihoodGradientFlags>(m, "TreeLikelihoodGradientFlags")
	    .value("TREE_HEIGHT", TreeLikelihoodGradientFlags::TREE_HEIGHT)
	    .value("SITE_MODEL", TreeLikelihoodGradientFlags::SITE_MODEL)
	    .value("SUBSTITUTION_MODEL", TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL)
	    .value("SUBSTITUTION_MODEL_RATES", TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL_RATES)
	    .value("SUBSTITUTION_MODEL_FREQUENCIES", TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL_FREQUENCIES)
	    .value("BRANCH_MODEL", TreeLikelihoodGradientFlags::BRANCH_MODEL);
	py::enum_<TreeTransformFlags>(m, "TreeTransformFlags")
	    .value("RATIO", TreeTransformFlags::RATIO)
	    .value("SHIFT", TreeTransformFlags::SHIFT)
	    .value("PROPORTION", TreeTransformFlags::PROPORTION);

	py::class_<DataTypeInterface>(m, "DataTypeInterface");
	py::class_<NucleotideDataTypeInterface, DataTypeInterface>(m, "NucleotideDataTypeInterface").def(py::init<>());
	py::class_<GeneralDataTypeInterface, DataTypeInterface>(m, "GeneralDataTypeInterface")
	    .def(py::init<const std::vector<std::string> &, std::optional<const std::map<std::string, std::vector<std::string>>>>(), py::arg("states"),
	         py::arg("ambiguities") = py::none());

	py::class_<ModelInterface>(m, "ModelInterface")
	    .def_readonly("parameter_count", &ModelInterface::parameterCount_)
	    .def("set_parameters", [](ModelInterface &self, darray p) { self.SetParameters(p.data()); })
	    .def("get_parameters", [](ModelInterface &self) {
		    std::vector<double> v(self.parameterCount_);
		    self.GetParameters(v.data());
		    return vec(v);
	    });
	py::class_<CallableModelInterface, ModelInterface>(m, "CallableModelInterface")
	    .def_readonly("gradient_length", &CallableModelInterface::gradientLength_)
	    .def("log_likelihood", &CallableModelInterface::LogLikelihood)
	    .def("gradient", [](CallableModelInterface &self) {
		    std::vector<double> g(self.gradientLength_);
		    self.Gradient(g.data());
		    return vec(g);
	    });

	py::class_<TreeModelInterface, ModelInterface>(m, "TreeModelInterface")
	    .def("get_node_count", &TreeModelInterface::GetNodeCount)
	    .def("get_tip_count", &TreeModelInterface::GetTipCount)
	    .def_readonly("node_map", &TreeModelInterface::nodeMap_)
	    .def("describe", [](TreeModelInterface &self) {  // node table, for checking the id conventions against fixtures
		    const phyamd::Tree &t = *self.GetTree();
		    py::dict d;
		    d["left"] = t.left;
		    d["right"] = t.right;
		    d["parent"] = t.parent;
		    d["class_id"] = t.class_id;
		    d["name"] = t.name;
		    d["distance"] = t.distance;
		    d["height"] = t.height;
		    d["root"] = t.root;
		    d["lowers"] = t.lowers;
		    d["ratios"] = t.ratios;
		    d["postorder"] = t.postorder;
		    return d;
	    });
	py::class_<UnRootedTreeModelInterface, TreeModelInterface>(m, "UnRootedTreeModelInterface")
	    .def(py::init<const std::string &, const std::vector<std::string> &>());
	py::class_<TimeTreeModelInterface, TreeModelInterface>(m, "TimeTreeModelInterface")
	    .def(py::init<const std::string &, const std::vector<std::string> &, const std::vector<double>>())
	    .def("get_node_heights", [](TimeTreeModelInterface &self) {
		    std::vector<double> h(self.GetTipCount() - 1);
		    self.GetNodeHeights(h.data());
		    return vec(h);
	    });
	py::class_<ReparameterizedTimeTreeModelInterface, TimeTreeModelInterface>(m, "ReparameterizedTimeTreeModelInterface")
	    .def(py::init<const std::string &, const std::vector<std::string> &, const std::vector<double>, TreeTransformFlags>())
	    .def("gradient_transform_jvp",
	         [](ReparameterizedTimeTreeModelInterface &self, darray hg) {
		         std::vector<double> g(self.GetTipCount() - 1);
		         self.GradientTransformJVP(g.data(), hg.data());
		         return vec(g);
	         })
	    .def("gradient_transform_jvp_with_heights",
	         [](ReparameterizedTimeTreeModelInterface &self, darray hg, darray heights) {
		         std::vector<double> g(self.GetTipCount() - 1);
		         self.GradientTransformJVP(g.data(), hg.data(), heights.data());
		         return vec(g);
	         })
	    .def("gradient_transform_jacobian",
	         [](ReparameterizedTimeTreeModelInterface &self) {
		         std::vector<double> g(self.GetTipCount() - 1);
		         self.GradientTransformJacobian(g.data());
		         return vec(g);
	         })
	    .def("transform_jacobian", &ReparameterizedTimeTreeModelInterface::TransformJacobian);

	py::class_<BranchModelInterface, ModelInterface>(m, "BranchModelInterface").def("set_rates", [](BranchModelInterface &self, darray r) {
		self.SetRates(r.data());
	});
	py::class_<StrictClockModelInterface, BranchModelInterface>(m, "StrictClockModelInterface")
	    .def(py::init<double, TreeModelInterface *>(), py::keep_alive<1, 3>())
	    .def("set_rate", &StrictClockModelInterface::SetRate);
	py::class_<SimpleClockModelInterface, BranchModelInterface>(m, "SimpleClockModelInterface")
	    .def(py::init<const std::vector<double> &, TreeModelInterface *>(), py::keep_alive<1, 3>());

	py::class_<SubstitutionModelInterface, ModelInterface>(m, "SubstitutionModelInterface")
	    .def("transition_matrix",  // host-side P(t) / dP/dt of the current parameters: parity checks of the eigen system
	         [](SubstitutionModelInterface &self, double t, bool derivative) {
		         phyamd::SubstModel &sm = *self.GetModel();
		         std::vector<double> P((size_t)sm.S * sm.S);
		         sm.p_t(t, P.data(), derivative);
		         return darray({sm.S, sm.S}, P.data());
	         },
	         py::arg("t"), py::arg("derivative") = false)
	    .def("rate_matrix_derivatives",  // dQ/dtheta [count][S][S] as uploaded for the substitution-model gradient
	         [](SubstitutionModelInterface &self, bool rates, bool frequencies) {
		         phyamd::SubstModel &sm = *self.GetModel();
		         std::vector<double> dQ;
		         sm.rate_matrix_derivatives(rates, frequencies, dQ);
		         return darray({(py::ssize_t)(dQ.size() / ((size_t)sm.S * sm.S)), (py::ssize_t)sm.S, (py::ssize_t)sm.S}, dQ.data());
	         },
	         py::arg("rates") = true, py::arg("frequencies") = true)
	    .def("eigen_system", [](SubstitutionModelInterface &self) {
		    phyamd::SubstModel &sm = *self.GetModel();
		    sm.update();
		    return py::make_tuple(vec(sm.eval), darray({sm.S, sm.S}, sm.evec.data()), darray({sm.S, sm.S}, sm.ivec.data()),
		                          darray({sm.S, sm.S}, sm.Q.data()));
	    });
	py::class_<JC69Interface, SubstitutionModelInterface>(m, "JC69Interface").def(py::init<>());
	py::class_<HKYInterface, SubstitutionModelInterface>(m, "HKYInterface")
	    .def(py::init<double, const std::vector<double> &>())
	    .def("set_kappa", &HKYInterface::SetKappa)
	    .def("set_frequencies", [](HKYInterface &self, darray f) { self.SetFrequencies(f.data()); });
	py::class_<GTRInterface, SubstitutionModelInterface>(m, "GTRInterface")
	    .def(py::init<const std::vector<double> &, const std::vector<double> &>())
	    .def("set_rates", [](GTRInterface &self, darray r) { self.SetRates(r.data()); })
	    .def("set_frequencies", [](GTRInterface &self, darray f) { self.SetFrequencies(f.data()); });
	py::class_<GeneralSubstitutionModelInterface, SubstitutionModelInterface>(m, "GeneralSubstitutionModelInterface")
	    .def(py::init<DataTypeInterface *, const std::vector<double> &, const std::vector<double> &, const std::vector<unsigned> &, bool>(),
	         py::keep_alive<1, 2>())
	    .def("set_rates", [](GeneralSubstitutionModelInterface &self, darray r) { self.SetRates(r.data()); })
	    .def("set_frequencies", [](GeneralSubstitutionModelInterface &self, darray f) { self.SetFrequencies(f.data()); });

	py::class_<SiteModelInterface, ModelInterface>(m, "SiteModelInterface")
	    .def("set_mu", &SiteModelInterface::SetMu)
	    .def("rates", [](SiteModelInterface &self) {
		    self.GetModel()->update();
		    std::vector<double> r(self.GetModel()->cat_count);
		    self.GetRates(r.data());
		    return vec(r);
	    })
	    .def("proportions", [](SiteModelInterface &self) {
		    self.GetModel()->update();
		    std::vector<double> r(self.GetModel()->cat_count);
		    self.GetProportions(r.data());
		    return vec(r);
	    });
	py::class_<ConstantSiteModelInterface, SiteModelInterface>(m, "ConstantSiteModelInterface")
	    .def(py::init<std::optional<double>>(), py::arg("mu") = py::none());
	py::class_<InvariantSiteModelInterface, SiteModelInterface>(m, "InvariantSiteModelInterface")
	    .def(py::init<double, std::optional<double>>(), py::arg("proportion_invariant"), py::arg("mu") = py::none())
	    .def("set_proportion_invariant", &InvariantSiteModelInterface::SetProportionInvariant);
	py::class_<DiscretizedSiteModelInterface, SiteModelInterface>(m, "DiscretizedSiteModelInterface")
	    .def("set_parameter", &DiscretizedSiteModelInterface::SetParameter)
	    .def("set_proportion_invariant", &DiscretizedSiteModelInterface::SetProportionInvariant)
	    .def("get_category_count", &DiscretizedSiteModelInterface::GetCategoryCount);
	py::class_<WeibullSiteModelInterface, DiscretizedSiteModelInterface>(m, "WeibullSiteModelInterface")
	    .def(py::init<double, size_t, std::optional<double>, std::optional<double>>(), py::arg("shape"), py::arg("categories"),
	         py::arg("proportion_invariant") = py::none(), py::arg("mu") = py::none())
	    .def("set_shape", &WeibullSiteModelInterface::SetShape);
	py::class_<GammaSiteModelInterface, DiscretizedSiteModelInterface>(m, "GammaSiteModelInterface")
	    .def(py::init<double, size_t, std::optional<double>, std::optional<double>>(), py::arg("shape"), py::arg("categories"),
	         py::arg("proportion_invariant") = py::none(), py::arg("mu") = py::none())
	    .def("set_shape", &GammaSiteModelInterface::SetShape)
	    .def("set_epsilon", &GammaSiteModelInterface::SetEpsilon);

	py::class_<TreeLikelihoodInterface, CallableModelInterface>(m, "TreeLikelihoodInterface")
	    .def(py::init<const std::vector<std::pair<std::string, std::string>> &, TreeModelInterface *, SubstitutionModelInterface *,
	                  SiteModelInterface *, std::optional<BranchModelInterface *>, bool, bool, bool>(),
	         py::arg("alignment"), py::arg("tree_model"), py::arg("substitution_model"), py::arg("site_model"), py::arg("branch_model") = py::none(),
	         py::arg("use_ambiguities") = false, py::arg("use_tip_states") = false, py::arg("include_jacobian") = false, py::keep_alive<1, 3>(),
	         py::keep_alive<1, 4>(), py::keep_alive<1, 5>(), py::keep_alive<1, 6>())
	    .def(py::init<const std::vector<std::string> &, const std::vector<std::string> &, TreeModelInterface *, SubstitutionModelInterface *,
	                  SiteModelInterface *, std::optional<BranchModelInterface *>, bool, bool, bool>(),
	         py::arg("taxa"), py::arg("attributes"), py::arg("tree_model"), py::arg("substitution_model"), py::arg("site_model"),
	         py::arg("branch_model") = py::none(), py::arg("use_ambiguities") = false, py::arg("use_tip_states") = false,
	         py::arg("include_jacobian") = false, py::keep_alive<1, 4>(), py::keep_alive<1, 5>(), py::keep_alive<1, 6>(), py::keep_alive<1, 7>())
	    .def("request_gradient", &TreeLikelihoodInterface::RequestGradient, py::arg("flags") = std::vector<TreeLikelihoodGradientFlags>())
	    .def("set_reference_compatibility", &TreeLikelihoodInterface::SetReferenceCompatibility)
	    .def("get_pattern_count", &TreeLikelihoodInterface::GetPatternCount)
	    .def("pattern_weights", [](TreeLikelihoodInterface &self) { return vec(self.PatternWeights()); })
	    .def("pattern_states", [](TreeLikelihoodInterface &self) {
		    const auto &s = self.PatternStates();
		    const py::ssize_t P = (py::ssize_t)self.GetPatternCount();
		    return py::array_t<unsigned char>({(py::ssize_t)(s.size() / (size_t)P), P}, s.data());
	    });

	// --- host-only helpers (CPU tests of the host logic; no device involved) ---
	m.def("compress_patterns", [](const std::string &datatype, const std::vector<std::string> &names, const std::vector<std::string> &seqs) {
		phyamd::DataType dt;
		if (datatype == "nucleotide") dt.kind = phyamd::DataTypeKind::Nucleotide, dt.state_count = 4;
		else if (datatype == "aa") dt.kind = phyamd::DataTypeKind::AminoAcid, dt.state_count = 20;
		else if (datatype == "codon") dt.kind = phyamd::DataTypeKind::Codon, dt.state_count = 61, dt.symbol_length = 3;
		else throw phyamd::Error("unknown datatype " + datatype);
		phyamd::Patterns p = phyamd::compress_patterns(dt, names, seqs);
		return py::make_tuple(py::array_t<unsigned char>({(py::ssize_t)p.taxon_count, (py::ssize_t)p.pattern_count}, p.states.data()), vec(p.weights));
	});
	m.def("compress_patterns_device", [](const std::string &datatype, const std::vector<std::string> &names, const std::vector<std::string> &seqs) {
		phyamd::DataType dt;
		if (datatype == "nucleotide") dt.kind = phyamd::DataTypeKind::Nucleotide, dt.state_count = 4;
		else if (datatype == "aa") dt.kind = phyamd::DataTypeKind::AminoAcid, dt.state_count = 20;
		else if (datatype == "codon") dt.kind = phyamd::DataTypeKind::Codon, dt.state_count = 61, dt.symbol_length = 3;
		else throw phyamd::Error("unknown datatype " + datatype);
		phyamd::Patterns p = phyamd::compress_patterns_device(dt, names, seqs);
		return py::make_tuple(py::array_t<unsigned char>({(py::ssize_t)p.taxon_count, (py::ssize_t)p.pattern_count}, p.states.data()), vec(p.weights));
	});
	// host substitution models by name (JC69, HKY, GTR, WAG, LG, MG94): normalised Q and its eigen system, as uploaded with
	// phyamd_set_eigen / phyamd_set_frequencies.  frequencies = None: the model's own (WAG) or uniform.
	m.def("substitution_model", [](const std::string &name, const std::vector<double> &rates, std::optional<std::vector<double>> frequencies) {
		phyamd::SubstModel sm;
		sm.name = name;
		sm.S = (name == "WAG" || name == "LG") ? 20 : name == "MG94" ? 61 : 4;
		sm.rates = rates;
		if (frequencies) sm.freqs = *frequencies;
		else if (name == "WAG") sm.freqs.assign(phyamd::wag_frequencies(), phyamd::wag_frequencies() + 20);
		else sm.freqs.assign(sm.S, 1.0 / sm.S);
		sm.update();
		const ssize_t S = sm.S;
		py::dict d;
		d["state_count"] = sm.S;
		d["frequencies"] = vec(sm.freqs);
		d["Q"] = darray({S, S}, sm.Q.data());
		d["eval"] = vec(sm.eval);
		d["evec"] = darray({S, S}, sm.evec.data());
		d["ivec"] = darray({S, S}, sm.ivec.data());
		return d;
	}, py::arg("name"), py::arg("rates") = std::vector<double>(), py::arg("frequencies") = std::nullopt);
	m.def("gamma_quantile", &phyamd::gamma_quantile);
	m.def("reg_lower_gamma", &phyamd::reg_lower_gamma);
}
