// Sanitizer driver for the host-side model code (tree.cpp, patterns.cpp, models.cpp): no GPU, no engine.
// Built and run by tests/test_host_sanitizers.py with -fsanitize=address,undefined; exits non-zero on any finding
// (ASan/UBSan abort) or on a failed self-check.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "phyamd_host.hpp"

using namespace phyamd;

static int fails = 0;
#define CHECK(c)                                                  \
	do {                                                          \
		if (!(c)) {                                               \
			std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
			fails++;                                              \
		}                                                         \
	} while (0)

static void read_fasta(const std::string &path, std::vector<std::string> &names, std::vector<std::string> &seqs) {
	std::ifstream f(path);
	std::string line;
	while (std::getline(f, line)) {
		if (line.empty()) continue;
		if (line[0] == '>') {
			names.push_back(line.substr(1));
			seqs.emplace_back();
		} else
			seqs.back() += line;
	}
}

int main(int argc, char **argv) {
	if (argc < 3) {
		std::fprintf(stderr, "usage: host_sanitize <aln.fa> <tree.nwk>\n");
		return 2;
	}
	std::vector<std::string> names, seqs;
	read_fasta(argv[1], names, seqs);
	std::ifstream tf(argv[2]);
	std::stringstream ss;
	ss << tf.rdbuf();
	std::string newick = ss.str();
	while (!newick.empty() && (newick.back() == '\n' || newick.back() == '\r')) newick.pop_back();
	CHECK(!names.empty());

	// patterns: compression, ragged input, unknown symbols
	DataType nuc;
	Patterns p = compress_patterns(nuc, names, seqs);
	CHECK(p.pattern_count > 0 && p.pattern_count <= p.site_count);
	double wsum = 0;
	for (double w : p.weights) wsum += w;
	CHECK(std::fabs(wsum - p.site_count) < 1e-9);
	try {
		std::vector<std::string> bad = seqs;
		bad[0].pop_back();
		(void)compress_patterns(nuc, names, bad);
		CHECK(!"ragged alignment accepted");
	} catch (const Error &) {
	}
	DataType gen;
	gen.kind = DataTypeKind::General;
	gen.state_count = 3;
	gen.symbol_length = 1;
	gen.states = {"a", "b", "c"};
	gen.ambiguities.emplace_back("ab", std::vector<int>{0, 1});
	CHECK(gen.encode_string("b") == 1 && gen.encode_string("ab") == 3 && gen.encode_string("zzz") == 4);
	double part[3];
	gen.partial(3, part);
	CHECK(part[0] == 1 && part[1] == 1 && part[2] == 0);
	gen.partial(4, part);
	CHECK(part[0] == 1 && part[2] == 1);

	// trees: unrooted, polytomy, quoted names, time tree + ratio transform round trip
	Tree t = make_unrooted_tree(newick, names);
	CHECK(t.tip_count == (int)names.size() && t.node_count == 2 * t.tip_count - 1);
	CHECK((int)t.postorder.size() == t.node_count);
	{
		std::vector<std::string> tx = {"A", "B x", "C", "D"};
		Tree q = parse_newick("(A:0.1,'B x':0.2,(C:0.3,D:0.4):0.5);", tx, true);
		CHECK(q.tip_count == 4 && q.node_count == 7);
		try {
			(void)parse_newick("(A:0.1,(B:0.2", tx, true);
			CHECK(!"truncated newick accepted");
		} catch (const Error &) {
		}
		std::vector<double> dates = {0.0, 1.0, 2.0, 0.5};
		Tree tt = make_time_tree("((A:2,'B x':1):1,(C:0.5,D:2):0.5);", tx, dates);
		enable_ratio_transform(tt);
		std::vector<double> h0 = tt.height;
		heights_from_ratios(tt);
		for (int n = 0; n < tt.node_count; n++) CHECK(std::fabs(tt.height[n] - h0[n]) < 1e-12);
		std::vector<double> hg(tt.tip_count - 1, 1.0), g(tt.tip_count - 1, 0.0);
		ratio_transform_jvp(tt, hg.data(), g.data());
		ratio_transform_log_jacobian_gradient(tt, g.data());
		CHECK(std::isfinite(ratio_transform_log_jacobian(tt)));
	}

	// substitution models: Q, eigen system, derivatives; site models
	for (const char *name : {"JC69", "HKY", "GTR", "GENERAL"}) {
		SubstModel m;
		m.name = name;
		m.freqs = {0.3, 0.2, 0.2, 0.3};
		if (m.name == "HKY") m.rates = {2.5};
		if (m.name == "GTR") m.rates = {1.2, 3.1, 0.7, 0.9, 2.8};
		if (m.name == "GENERAL") {
			m.rates = {1.0, 2.0};
			m.structure = {0, 1, 0, 0, 1, 0};
		}
		if (m.name == "JC69") m.freqs = {0.25, 0.25, 0.25, 0.25};
		m.update();
		double P[16];
		m.p_t(0.37, P);
		for (int i = 0; i < 4; i++) {
			double row = 0;
			for (int j = 0; j < 4; j++) row += P[i * 4 + j];
			CHECK(std::fabs(row - 1.0) < 1e-12);
		}
		std::vector<double> dQ;
		m.rate_matrix_derivatives(true, true, dQ);
		CHECK(dQ.size() == (size_t)(m.rate_parameter_count() + (m.name == "JC69" ? 0 : 4)) * 16);
	}
	{
		SubstModel big;  // 20 states through the Jacobi solver
		big.S = 20;
		big.name = "GENERAL";
		big.rates = {1.0, 0.5, 2.0};
		big.structure.resize(190);
		for (size_t i = 0; i < 190; i++) big.structure[i] = (unsigned)(i % 3);
		big.freqs.assign(20, 0.05);
		big.update();
		std::vector<double> P(400);
		big.p_t(0.2, P.data());
		double row = 0;
		for (int j = 0; j < 20; j++) row += P[j];
		CHECK(std::fabs(row - 1.0) < 1e-10);
	}
	for (RateDistribution d : {RateDistribution::Gamma, RateDistribution::Weibull}) {
		SiteModel sm;
		sm.dist = d;
		sm.cat_count = 5;
		sm.shape = 0.7;
		sm.has_pinv = true;
		sm.pinv = 0.2;
		sm.update();
		double mean = 0;
		for (int c = 0; c < sm.cat_count; c++) mean += sm.cat_rates[c] * sm.cat_props[c];
		CHECK(std::fabs(mean - 1.0) < 1e-9);
		std::vector<double> in(sm.cat_count, 0.3);
		CHECK(std::isfinite(sm.shape_gradient(in.data())) && std::isfinite(sm.pinv_gradient(in.data())));
	}
	CHECK(std::fabs(reg_lower_gamma(2.0, gamma_quantile(0.3, 2.0, 1.0)) - 0.3) < 1e-9);
	if (fails) return 1;
	std::puts("host_sanitize: ok");
	return 0;
}
