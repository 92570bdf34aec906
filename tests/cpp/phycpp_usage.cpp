// A caller written against the reference's wrapper header (src/phycpp/physher.hpp): same class names, constructors and
// methods -- only the #include changes.  tests/test_phycpp_source_compat.py compiles and links it on the CPU
// (source compatibility) and runs it on the GPU box against a golden case.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#ifdef PHYCPP_REFERENCE
#include "phycpp/physher.hpp"  // the reference's own wrapper (oracle/Makefile: phycpp_usage_ref, run on the device through seam A)
#else
#include "phycpp_amd/physher.hpp"
#endif

static void read_fasta(const std::string &path, std::vector<std::pair<std::string, std::string>> &aln) {
	std::ifstream f(path);
	std::string line;
	while (std::getline(f, line)) {
		if (line.empty()) continue;
		if (line[0] == '>') aln.emplace_back(line.substr(1), "");
		else aln.back().second += line;
	}
}

int main(int argc, char **argv) {
	if (argc < 3) {
		std::fprintf(stderr, "usage: phycpp_usage <aln.fa> <tree.nwk>\n");
		return 2;
	}
	std::vector<std::pair<std::string, std::string>> alignment;
	read_fasta(argv[1], alignment);
	std::ifstream tf(argv[2]);
	std::stringstream ss;
	ss << tf.rdbuf();
	std::string newick = ss.str();
	while (!newick.empty() && (newick.back() == '\n' || newick.back() == '\r')) newick.pop_back();
	std::vector<std::string> taxa;
	for (const auto &kv : alignment) taxa.push_back(kv.first);

	UnRootedTreeModelInterface tree(newick, taxa);
	GTRInterface subst({1.2, 3.1, 0.7, 0.9, 2.8}, {0.3, 0.2, 0.2, 0.3});
	GammaSiteModelInterface site(0.5, 4, std::nullopt, std::nullopt);
	TreeLikelihoodInterface tlk(alignment, &tree, &subst, &site, std::nullopt, false, false, false);

	const double lnl = tlk.LogLikelihood();
	tlk.RequestGradient({TreeLikelihoodGradientFlags::TREE_HEIGHT, TreeLikelihoodGradientFlags::SITE_MODEL,
	                     TreeLikelihoodGradientFlags::SUBSTITUTION_MODEL});
	std::vector<double> gradient(tlk.gradientLength_);
	tlk.Gradient(gradient.data());
	std::printf("lnL %.12f\ngradient_length %zu\n", lnl, (size_t)tlk.gradientLength_);
	for (double g : gradient) std::printf("%.12g\n", g);

	// parameter updates through the reference's setters
	std::vector<double> bl(tree.GetNodeCount() - 2);
	tree.GetParameters(bl.data());
	bl[0] *= 1.5;
	tree.SetParameters(bl.data());
	subst.SetRates(std::vector<double>{1.0, 2.0, 1.0, 1.0, 2.0}.data());
	site.SetShape(0.8);
	std::printf("lnL2 %.12f\n", tlk.LogLikelihood());
	return 0;
}
