// tree.cpp -- Newick parsing with physher's node-id conventions and the node-height ratio transform.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <functional>

#include "phyamd_host.hpp"

namespace phyamd {

namespace {

constexpr double BL_MIN = 1.0e-8;  // node.h:26

struct RawNode {
	int left = -1, right = -1, parent = -1;
	std::string name;
	double distance = 0.0;
	bool has_length = false;
	bool poly = false;  // inserted to resolve a polytomy
};

struct Parser {
	const std::string &s;
	size_t i = 0;
	std::vector<RawNode> nodes;
	bool contain_bl;

	Parser(const std::string &str, bool bl) : s(str), contain_bl(bl) {}

	int add(int parent) {
		nodes.emplace_back();
		nodes.back().parent = parent;
		return (int)nodes.size() - 1;
	}
	void skip_ws() {
		while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
	}
	// comment, label (ignored for clades) and ":length" after a name or a ')'  (tree.c:77-181)
	void description(int n) {
		skip_ws();
		if (i < s.size() && s[i] == '[') {
			while (i < s.size() && s[i] != ']') i++;
			if (i < s.size()) i++;
		}
		while (i < s.size() && s[i] != ':' && s[i] != ',' && s[i] != ')' && s[i] != ';') i++;
		if (i < s.size() && s[i] == ':') {
			i++;
			if (i < s.size() && s[i] == '[') {
				while (i < s.size() && s[i] != ']') i++;
				if (i < s.size()) i++;
			}
			size_t b = i;
			while (i < s.size() && s[i] != ',' && s[i] != ')' && s[i] != ';') i++;
			nodes[n].distance = std::atof(s.substr(b, i - b).c_str());
			nodes[n].has_length = true;
		}
		if (contain_bl) nodes[n].distance = std::max(BL_MIN, nodes[n].has_length ? nodes[n].distance : -INFINITY);
	}
	// attach child c under cur; a third child goes under a new node that takes over cur's right slot (tree.c:621-640, 736-756)
	void attach(int cur, int c, bool child_is_tip) {
		if (nodes[cur].left < 0) nodes[cur].left = c;
		else if (nodes[cur].right < 0) nodes[cur].right = c;
		else {
			const int t = add(cur);
			nodes[t].poly = true;
			nodes[t].distance = child_is_tip ? 0.0 : BL_MIN;
			const int r = nodes[cur].right;
			nodes[cur].right = t;
			nodes[t].left = r;
			nodes[t].right = c;
			nodes[r].parent = t;
			nodes[c].parent = t;
		}
	}
	int parse() {
		int depth = 0;
		for (char ch : s) depth += ch == '(' ? 1 : ch == ')' ? -1 : 0;
		if (depth != 0) throw Error("The newick tree is malformed: Number opening parenthesis != number closing parenthesis");
		skip_ws();
		if (i >= s.size() || s[i] != '(') throw Error("newick string must start with '('");
		const int root = add(-1);
		int cur = root;
		i++;
		while (i < s.size()) {
			const char ch = s[i];
			if (ch == '(') {
				const int n = add(cur);
				attach(cur, n, false);
				cur = n;
				i++;
			} else if (ch == ',' || std::isspace((unsigned char)ch)) {
				i++;
			} else if (ch == ')') {
				i++;
				if (cur == root) break;  // anything after the last ')' (root label / length / ';') is ignored
				description(cur);
				cur = nodes[cur].parent;
				if (nodes[cur].poly) cur = nodes[cur].parent;
			} else if (ch == ';') {
				break;
			} else {  // a tip name
				size_t b = i;
				while (i < s.size() && s[i] != ':' && s[i] != ',' && s[i] != ')' && s[i] != '[') i++;
				std::string nm = s.substr(b, i - b);
				while (!nm.empty() && std::isspace((unsigned char)nm.back())) nm.pop_back();
				if (nm.size() >= 2 && ((nm.front() == '\'' && nm.back() == '\'') || (nm.front() == '"' && nm.back() == '"'))) nm = nm.substr(1, nm.size() - 2);
				if (nm.empty()) throw Error("empty taxon name in newick string");
				const int n = add(cur);
				nodes[n].name = nm;
				description(n);
				attach(cur, n, true);
			}
		}
		if (cur != root) throw Error("newick string ended inside a clade");
		return root;
	}
};

}  // namespace

Tree parse_newick(const std::string &newick, const std::vector<std::string> &taxa, bool contain_bl) {
	Parser p(newick, contain_bl);
	const int raw_root = p.parse();
	const auto &rn = p.nodes;
	for (const auto &n : rn)
		if ((n.left < 0) != (n.right < 0)) throw Error("newick clade with a single child");
	int tips = 0;
	for (const auto &n : rn) tips += n.left < 0;
	if (tips != (int)taxa.size()) throw Error("newick has " + std::to_string(tips) + " tips but " + std::to_string(taxa.size()) + " taxa were given");
	if ((int)rn.size() != 2 * tips - 1) throw Error("tree is not binary after polytomy resolution");

	Tree t;
	t.tip_count = tips;
	t.node_count = (int)rn.size();
	const int N = t.node_count;
	t.left.assign(N, -1);
	t.right.assign(N, -1);
	t.parent.assign(N, -1);
	t.class_id.assign(N, -1);
	t.name.assign(N, "");
	t.distance.assign(N, 0.0);
	t.height.assign(N, 0.0);
	// post-order over the raw nodes: ids (tree.c:202-224)
	std::vector<int> raw_post;
	raw_post.reserve(N);
	{
		std::vector<std::pair<int, int>> st{{raw_root, 0}};
		while (!st.empty()) {
			auto &[n, state] = st.back();
			if (rn[n].left < 0) {
				raw_post.push_back(n);
				st.pop_back();
			} else if (state == 0) {
				state = 1;
				st.push_back({rn[n].left, 0});
			} else if (state == 1) {
				state = 2;
				st.push_back({rn[n].right, 0});
			} else {
				raw_post.push_back(n);
				st.pop_back();
			}
		}
	}
	std::vector<int> id_of(N, -1);
	int internals = 0;
	std::vector<char> taken(tips, 0);
	for (int r : raw_post) {
		if (rn[r].left < 0) {
			auto it = std::find(taxa.begin(), taxa.end(), rn[r].name);
			if (it == taxa.end()) throw Error("Could not find taxon " + rn[r].name + " in taxon list");
			const int idx = (int)(it - taxa.begin());
			if (taken[idx]) throw Error("taxon " + rn[r].name + " appears twice in the tree");
			taken[idx] = 1;
			id_of[r] = idx;
			t.class_id[idx] = idx;
		} else {
			id_of[r] = tips + internals;
			t.class_id[tips + internals] = internals;
			internals++;
		}
	}
	for (int r = 0; r < N; r++) {
		const int id = id_of[r];
		t.name[id] = rn[r].name;
		t.distance[id] = rn[r].distance;
		if (rn[r].left >= 0) {
			t.left[id] = id_of[rn[r].left];
			t.right[id] = id_of[rn[r].right];
			t.parent[t.left[id]] = id;
			t.parent[t.right[id]] = id;
		}
	}
	t.root = id_of[raw_root];
	t.distance[t.root] = 0.0;
	t.postorder.clear();
	for (int r : raw_post) t.postorder.push_back(id_of[r]);
	// pre-order: node, left subtree, right subtree
	t.preorder.clear();
	{
		std::vector<int> st{t.root};
		while (!st.empty()) {
			const int n = st.back();
			st.pop_back();
			t.preorder.push_back(n);
			if (t.left[n] >= 0) {
				st.push_back(t.right[n]);
				st.push_back(t.left[n]);
			}
		}
	}
	return t;
}

Tree make_unrooted_tree(const std::string &newick, const std::vector<std::string> &taxa) {
	Tree t = parse_newick(newick, taxa, true);
	const int rl = t.left[t.root], rr = t.right[t.root];
	if (t.distance[rr] != 0.0) {  // tree.c:1438-1443
		t.distance[rl] += t.distance[rr];
		t.distance[rr] = 0.0;
	}
	return t;
}

Tree make_time_tree(const std::string &newick, const std::vector<std::string> &taxa, const std::vector<double> &dates) {
	if (dates.size() != taxa.size()) throw Error("one date per taxon is required");
	Tree t = parse_newick(newick, taxa, false);
	t.time_mode = true;
	// init_dates2 (tree.c:394-424): heights are measured backwards from the most recent tip
	double mn = INFINITY, mx = -INFINITY;
	bool homochronous = true;
	for (int i = 0; i < t.tip_count; i++) {
		mn = std::min(mn, dates[i]);
		mx = std::max(mx, dates[i]);
		if (dates[i] != 0.0) homochronous = false;
		t.height[i] = dates[i];
	}
	if (!homochronous && mn != 0.0)
		for (int i = 0; i < t.tip_count; i++) t.height[i] = mx - dates[i];
	// init_heights_from_distances (tree.c:498-515, 536-547)
	for (int n : t.postorder) {
		if (t.is_leaf(n)) {
			if (homochronous) t.height[n] = 0.0;
			continue;
		}
		const int l = t.left[n], r = t.right[n];
		t.height[n] = std::max(t.height[l] + std::max(t.distance[l], 1.0e-6), t.height[r] + std::max(t.distance[r], 1.0e-6));
	}
	return t;
}

void enable_ratio_transform(Tree &t) {
	if (!t.time_mode) throw Error("the ratio transform needs a time tree");
	const int N = t.node_count;
	t.lowers.assign(N, 0.0);
	for (int n : t.postorder)  // tree_transform_collect_lowers (treetransform.c:240-254)
		t.lowers[n] = t.is_leaf(n) ? t.height[n] : std::max(t.lowers[t.left[n]], t.lowers[t.right[n]]);
	t.ratios.assign(t.tip_count - 1, 0.0);
	for (int n = t.tip_count; n < N; n++) {
		if (n == t.root) t.ratios[t.class_id[n]] = t.height[n];
		else t.ratios[t.class_id[n]] = (t.height[n] - t.lowers[n]) / (t.height[t.parent[n]] - t.lowers[n]);
	}
	t.reparameterized = true;
}

void heights_from_ratios(Tree &t) {
	for (int n : t.preorder) {
		if (t.is_leaf(n)) continue;
		const double s = t.ratios[t.class_id[n]];
		if (n == t.root) t.height[n] = s;
		else t.height[n] = t.lowers[n] + (t.height[t.parent[n]] - t.lowers[n]) * s;
	}
}

// h_n = L_n + (h_parent - L_n) r_n, h_root = r_root.  Reverse sweep: the adjoint of a height collects its own
// gradient plus r_c times the adjoint of every internal child; d/d r_n = adjoint_n (h_parent - L_n).
void ratio_transform_jvp(const Tree &t, const double *gh, double *gradient) {
	std::vector<double> adj(t.tip_count - 1, 0.0);
	for (int n : t.postorder) {
		if (t.is_leaf(n)) continue;
		const int c = t.class_id[n];
		double a = gh[c];
		for (int ch : {t.left[n], t.right[n]})
			if (!t.is_leaf(ch)) a += adj[t.class_id[ch]] * t.ratios[t.class_id[ch]];
		adj[c] = a;
		gradient[c] = n == t.root ? a : a * (t.height[t.parent[n]] - t.lowers[n]);
	}
}

double ratio_transform_log_jacobian(const Tree &t) {
	double s = 0.0;
	for (int n = t.tip_count; n < t.node_count; n++)
		if (n != t.root) s += std::log(t.height[t.parent[n]] - t.lowers[n]);
	return s;
}

void ratio_transform_log_jacobian_gradient(const Tree &t, double *gradient) {
	// log|J| depends on the heights of parents only: d/dh_p = sum over internal children n of 1 / (h_p - L_n)
	std::vector<double> gh(t.tip_count - 1, 0.0), g(t.tip_count - 1, 0.0);
	for (int n = t.tip_count; n < t.node_count; n++)
		if (n != t.root) gh[t.class_id[t.parent[n]]] += 1.0 / (t.height[t.parent[n]] - t.lowers[n]);
	ratio_transform_jvp(t, gh.data(), g.data());
	for (int i = 0; i < t.tip_count - 1; i++) gradient[i] += g[i];
}

}  // namespace phyamd
