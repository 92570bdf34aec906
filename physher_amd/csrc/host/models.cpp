// models.cpp -- substitution-model eigen systems and discrete-rate site models (host side, O(S^3) / O(C)).
#include <algorithm>
#include <cmath>
#include <limits>

#include "phyamd_host.hpp"

namespace phyamd {

// ---------------------------------------------------------------------------------------------
// substitution models
// ---------------------------------------------------------------------------------------------

namespace {
#include "aa_models.inc"

// the universal genetic code over the 64 triplets in the order AAA, AAC, AAG, AAT, ACA, ... (A < C < G < T, first position
// slowest): one letter per amino acid, '*' = stop.  The 61 sense codons, in this order, are the states of a codon model
// (sitepattern.c:808-819, geneticcode.h).
const char UNIVERSAL_CODE[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
}  // namespace

const double *wag_frequencies() { return WAG_FREQUENCIES; }

void build_symmetric_rates(const SubstModel &m, std::vector<double> &R) {
	const int S = m.S;
	R.assign((size_t)S * S, 0.0);
	auto set = [&](int i, int j, double v) { R[(size_t)i * S + j] = R[(size_t)j * S + i] = v; };
	if (m.name == "JC69") {
		for (int i = 0; i < S; i++)
			for (int j = i + 1; j < S; j++) set(i, j, 1.0);
	} else if (m.name == "HKY") {  // transitions A<->G and C<->T carry kappa
		if (m.rates.size() != 1 || S != 4) throw Error("HKY needs kappa and 4 states");
		for (int i = 0; i < 4; i++)
			for (int j = i + 1; j < 4; j++) set(i, j, (j - i == 2) ? m.rates[0] : 1.0);
	} else if (m.name == "GTR") {  // gtr.c:163-200: pairs in the order AC AG AT CG CT GT; five rates are relative to GT = 1
		if (S != 4 || (m.rates.size() != 5 && m.rates.size() != 6)) throw Error("GTR needs 5 (relative to GT) or 6 (simplex) rates");
		int idx = 0;
		for (int i = 0; i < 4; i++)
			for (int j = i + 1; j < 4; j++) {
				set(i, j, idx < (int)m.rates.size() ? m.rates[idx] : 1.0);
				idx++;
			}
	} else if (m.name == "WAG" || m.name == "LG") {  // _wag_update_Q / _lg_update_Q (wag.c:23-36, lg.c:23-36): tabulated exchangeabilities
		if (S != 20) throw Error(m.name + " is a 20-state model");
		const double *tab = m.name == "WAG" ? WAG_EXCHANGEABILITIES : LG_EXCHANGEABILITIES;
		size_t t = 0;
		for (int i = 0; i < 20; i++)
			for (int j = i + 1; j < 20; j++) set(i, j, tab[t++]);
	} else if (m.name == "MG94") {
		// _mg_update_Q (mg94.c:62-138): sense codons that differ at exactly one position exchange at
		// (kappa if that difference is a transition) x (alpha if synonymous, beta if not); rates = {kappa, alpha, beta}
		if (S != 61 || m.rates.size() != 3) throw Error("MG94 needs 61 states (universal code) and the rates kappa, alpha, beta");
		const double kappa = m.rates[0], alpha = m.rates[1], beta = m.rates[2];
		int sense[61], n = 0;
		for (int x = 0; x < 64; x++)
			if (UNIVERSAL_CODE[x] != '*') sense[n++] = x;
		for (int i = 0; i < 61; i++)
			for (int j = i + 1; j < 61; j++) {
				const int a = sense[i], b = sense[j];
				int differing = 0, na = 0, nb = 0;
				for (int pos = 0; pos < 3; pos++) {
					const int xa = (a >> (2 * (2 - pos))) & 3, xb = (b >> (2 * (2 - pos))) & 3;
					if (xa != xb) differing++, na = xa, nb = xb;
				}
				if (differing != 1) continue;
				const bool transition = (na ^ nb) == 2;  // A(0) <-> G(2), C(1) <-> T(3)
				const bool synonymous = UNIVERSAL_CODE[a] == UNIVERSAL_CODE[b];
				set(i, j, (transition ? kappa : 1.0) * (synonymous ? alpha : beta));
			}
	} else if (m.name == "GENERAL") {
		// structure maps each pair of states to a rate index.  Layouts: S(S-1)/2 = the upper triangle row by row (the packed
		// form of gensubst.c:177-183; the reference's reader of it, :130-151, indexes past the array); S(S-1) = upper then
		// lower triangle, both row by row (_nonreversible_update_Q, gensubst.c:60-79); S*S = full matrix.
		const size_t full = (size_t)S * S, tri = (size_t)S * (S - 1) / 2;
		const size_t n = m.structure.size();
		if (n != full && n != tri && !(n == 2 * tri && S > 1)) throw Error("GENERAL: structure must have S*(S-1)/2, S*(S-1) or S*S entries");
		size_t t = 0;
		for (int i = 0; i < S; i++)
			for (int j = i + 1; j < S; j++, t++) {
				unsigned a, mirrored;
				if (n == full) {
					a = m.structure[(size_t)i * S + j];
					mirrored = m.structure[(size_t)j * S + i];
				} else {
					a = m.structure[t];
					mirrored = n == tri ? a : m.structure[tri + (size_t)j * (j - 1) / 2 + i];  // row j of the lower triangle starts at j(j-1)/2
				}
				if (a != mirrored) throw Error("GENERAL: only reversible (symmetric) structures run on this path");
				if (a >= m.rates.size()) throw Error("GENERAL: structure refers to a missing rate");
				set(i, j, m.rates[a]);
			}
	} else
		throw Error("unknown substitution model " + m.name);
}

namespace {

// cyclic Jacobi for a symmetric matrix A (row-major, destroyed): eigenvalues w, eigenvectors in the columns of V
void jacobi_eigen(std::vector<double> &A, int n, std::vector<double> &w, std::vector<double> &V) {
	V.assign((size_t)n * n, 0.0);
	for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
	for (int sweep = 0; sweep < 100; sweep++) {
		double off = 0.0, diag = 0.0;
		for (int i = 0; i < n; i++) {
			diag += A[(size_t)i * n + i] * A[(size_t)i * n + i];
			for (int j = i + 1; j < n; j++) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
		}
		if (off <= 1e-34 * (diag + off) || off == 0.0) break;
		for (int p = 0; p < n - 1; p++)
			for (int q = p + 1; q < n; q++) {
				const double apq = A[(size_t)p * n + q];
				if (apq == 0.0) continue;
				const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
				const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
				for (int k = 0; k < n; k++) {  // columns p, q
					const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
					A[(size_t)k * n + p] = c * akp - s * akq;
					A[(size_t)k * n + q] = s * akp + c * akq;
				}
				for (int k = 0; k < n; k++) {  // rows p, q
					const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
					A[(size_t)p * n + k] = c * apk - s * aqk;
					A[(size_t)q * n + k] = s * apk + c * aqk;
				}
				for (int k = 0; k < n; k++) {
					const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
					V[(size_t)k * n + p] = c * vkp - s * vkq;
					V[(size_t)k * n + q] = s * vkp + c * vkq;
				}
			}
	}
	w.resize(n);
	for (int i = 0; i < n; i++) w[i] = A[(size_t)i * n + i];
}

}  // namespace

void SubstModel::update() {
	if (!dirty) return;
	if ((int)freqs.size() != S) throw Error("substitution model: need one frequency per state");
	std::vector<double> R;
	build_symmetric_rates(*this, R);
	Q.assign((size_t)S * S, 0.0);
	for (int i = 0; i < S; i++) {
		double row = 0.0;
		for (int j = 0; j < S; j++)
			if (i != j) {
				Q[(size_t)i * S + j] = R[(size_t)i * S + j] * freqs[j];  // gtr.c:172-176
				row += Q[(size_t)i * S + j];
			}
		Q[(size_t)i * S + i] = -row;  // make_zero_rows
	}
	norm = 1.0;
	if (normalize) {  // normalize_Q (substmodel.c:1135-1143): one expected substitution per unit time
		norm = 0.0;
		for (int i = 0; i < S; i++) norm -= Q[(size_t)i * S + i] * freqs[i];
		for (double &q : Q) q /= norm;
	}
	// reversible Q: D^{1/2} Q D^{-1/2} is symmetric; its orthonormal eigenvectors give U = D^{-1/2} V, U^-1 = V^T D^{1/2}
	std::vector<double> d(S), B((size_t)S * S), V;
	for (int i = 0; i < S; i++) {
		if (!(freqs[i] > 0.0)) throw Error("substitution model: frequencies must be positive");
		d[i] = std::sqrt(freqs[i]);
	}
	for (int i = 0; i < S; i++)
		for (int j = 0; j < S; j++) B[(size_t)i * S + j] = d[i] * Q[(size_t)i * S + j] / d[j];
	for (int i = 0; i < S; i++)
		for (int j = i + 1; j < S; j++) B[(size_t)i * S + j] = B[(size_t)j * S + i] = 0.5 * (B[(size_t)i * S + j] + B[(size_t)j * S + i]);
	jacobi_eigen(B, S, eval, V);
	evec.assign((size_t)S * S, 0.0);
	ivec.assign((size_t)S * S, 0.0);
	for (int i = 0; i < S; i++)
		for (int k = 0; k < S; k++) {
			evec[(size_t)i * S + k] = V[(size_t)i * S + k] / d[i];
			ivec[(size_t)k * S + i] = V[(size_t)i * S + k] * d[i];
		}
	dirty = false;
}

void SubstModel::rate_matrix_derivatives(bool want_rates, bool want_freqs, std::vector<double> &dQ) {
	update();
	dQ.clear();
	if (name == "JC69") return;  // no dPdp in the reference (jc69.c): nothing to differentiate
	std::vector<double> R;
	build_symmetric_rates(*this, R);
	std::vector<double> dR((size_t)S * S), dF(S), dq((size_t)S * S);
	// one parameter: exchangeability perturbation dR (symmetric) and frequency perturbation dF
	auto emit = [&]() {
		double dnorm = 0.0;
		for (int i = 0; i < S; i++) {
			double row = 0.0;
			for (int j = 0; j < S; j++)
				if (i != j) {
					dq[(size_t)i * S + j] = dR[(size_t)i * S + j] * freqs[j] + R[(size_t)i * S + j] * dF[j];  // build_Q_flat (substmodel.c:450-466)
					row += dq[(size_t)i * S + j];
				}
			dq[(size_t)i * S + i] = -row;
			// d norm = -sum_i ( dQ^_ii pi_i + Q^_ii dpi_i ), Q^_ii = Q_ii norm  (gtr.c:314-316)
			dnorm -= dq[(size_t)i * S + i] * freqs[i] + Q[(size_t)i * S + i] * norm * dF[i];
		}
		for (size_t a = 0; a < dq.size(); a++) dQ.push_back(normalize ? (dq[a] - Q[a] * dnorm) / norm : dq[a]);  // gtr.c:319-324
	};
	if (want_rates) {
		std::fill(dF.begin(), dF.end(), 0.0);
		const std::vector<double> saved = rates;
		for (size_t r = 0; r < saved.size(); r++) {
			// R is linear in every rate: dR/dr = R(e_r) - R(0), with the constants (GTR's GT = 1, HKY's transversions) cancelling
			std::vector<double> R1, R0;
			std::fill(rates.begin(), rates.end(), 0.0);
			build_symmetric_rates(*this, R0);
			rates[r] = 1.0;
			build_symmetric_rates(*this, R1);
			for (size_t a = 0; a < dR.size(); a++) dR[a] = R1[a] - R0[a];
			rates = saved;
			emit();
		}
	}
	if (want_freqs) {
		std::fill(dR.begin(), dR.end(), 0.0);
		for (int f = 0; f < S; f++) {
			std::fill(dF.begin(), dF.end(), 0.0);
			dF[f] = 1.0;
			emit();
		}
	}
}

void SubstModel::p_t(double t, double *P, bool derivative) {
	update();
	for (int i = 0; i < S; i++)
		for (int j = 0; j < S; j++) {
			double s = 0.0;
			for (int k = 0; k < S; k++) {
				const double e = std::exp(eval[k] * t);
				s += evec[(size_t)i * S + k] * (derivative ? eval[k] * e : e) * ivec[(size_t)k * S + j];
			}
			P[(size_t)i * S + j] = derivative ? s : std::fabs(s);  // substmodel.c:552
		}
}

// ---------------------------------------------------------------------------------------------
// incomplete gamma function and its inverse
// ---------------------------------------------------------------------------------------------

double reg_lower_gamma(double a, double x) {
	if (x <= 0.0) return 0.0;
	const double lg = std::lgamma(a);
	if (x < a + 1.0) {  // series
		double term = 1.0 / a, sum = term, ap = a;
		for (int n = 0; n < 10000; n++) {
			ap += 1.0;
			term *= x / ap;
			sum += term;
			if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
		}
		return sum * std::exp(-x + a * std::log(x) - lg);
	}
	// continued fraction for Q(a, x) (modified Lentz)
	const double tiny = 1e-300;
	double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
	for (int i = 1; i < 10000; i++) {
		const double an = -i * (i - a);
		b += 2.0;
		d = an * d + b;
		if (std::fabs(d) < tiny) d = tiny;
		c = b + an / c;
		if (std::fabs(c) < tiny) c = tiny;
		d = 1.0 / d;
		const double del = d * c;
		h *= del;
		if (std::fabs(del - 1.0) < 1e-16) break;
	}
	return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}

double gamma_quantile(double p, double shape, double rate) {
	if (!(shape > 0.0) || !(rate > 0.0) || p < 0.0 || p >= 1.0) return std::numeric_limits<double>::quiet_NaN();
	if (p == 0.0) return 0.0;
	const double a = shape, lg = std::lgamma(a);
	// starting point: Wilson-Hilferty for a > 1, small-x asymptote otherwise
	double x;
	if (a > 1.0) {
		const double pp = p < 0.5 ? p : 1.0 - p, t = std::sqrt(-2.0 * std::log(pp));
		double z = (2.30753 + t * 0.27061) / (1.0 + t * (0.99229 + t * 0.04481)) - t;
		if (p < 0.5) z = -z;
		x = std::max(1e-3, a * std::pow(1.0 - 1.0 / (9.0 * a) - z / (3.0 * std::sqrt(a)), 3));
	} else {
		const double t = 1.0 - a * (0.253 + a * 0.12);
		x = p < t ? std::pow(p / t, 1.0 / a) : 1.0 - std::log(1.0 - (p - t) / (1.0 - t));
	}
	double lo = 0.0, hi = std::numeric_limits<double>::infinity();
	for (int it = 0; it < 200; it++) {
		const double err = reg_lower_gamma(a, x) - p;
		if (err > 0) hi = std::min(hi, x);
		else lo = std::max(lo, x);
		const double dens = std::exp(-x + (a - 1.0) * std::log(x) - lg);
		double step;
		if (dens > 0.0 && std::isfinite(dens)) {
			const double u = err / dens;
			step = u / (1.0 - 0.5 * std::min(1.0, u * ((a - 1.0) / x - 1.0)));  // Halley
		} else
			step = 0.0;
		double nx = x - step;
		if (!(nx > lo) || !(nx < hi) || step == 0.0) nx = std::isfinite(hi) ? 0.5 * (lo + hi) : 2.0 * x + 1e-300;  // bisection safeguard
		if (std::fabs(nx - x) <= 4e-16 * nx) {
			x = nx;
			break;
		}
		x = nx;
	}
	return x / rate;
}

// ---------------------------------------------------------------------------------------------
// site models
// ---------------------------------------------------------------------------------------------

void SiteModel::update() {
	if (!dirty) return;
	cat_rates.assign(cat_count, 1.0);
	cat_props.assign(cat_count, 1.0 / cat_count);
	const int inv = has_pinv ? 1 : 0;
	const int n = cat_count - inv;  // variable categories
	if (dist == RateDistribution::Constant) {
		if (has_pinv) {  // +I (sitemodel.c:641-650)
			if (cat_count != 2) throw Error("+I alone has exactly two categories");
			cat_props = {pinv, 1.0 - pinv};
			cat_rates = {0.0, 1.0 / (1.0 - pinv)};
		}
		dirty = false;
		return;
	}
	if (n < 1) throw Error("site model needs at least one variable category");
	const double variable = has_pinv ? 1.0 - pinv : 1.0;
	if (has_pinv) {
		cat_rates[0] = 0.0;
		cat_props[0] = pinv;
	}
	double mean = 0.0;
	for (int i = 0; i < n; i++) {  // median of each equal-probability class (sitemodel.c:660-700)
		const double q = (2.0 * i + 1.0) / (2.0 * n);
		double r;
		if (dist == RateDistribution::Gamma) r = gamma_quantile(q, shape, shape);
		else r = std::pow(-std::log(1.0 - q), 1.0 / shape);  // Weibull with lambda = 1
		cat_rates[i + inv] = r;
		cat_props[i + inv] = variable / n;
		mean += r;
	}
	mean = variable * mean / n;  // sitemodel.c:735-741: the mixture (including the invariant class) has mean 1
	for (int i = 0; i < n; i++) cat_rates[i + inv] /= mean;
	dirty = false;
}

double SiteModel::quantile(double p, double a) const {
	return dist == RateDistribution::Gamma ? gamma_quantile(p, a, a) : std::pow(-std::log(1.0 - p), 1.0 / a);
}

double SiteModel::shape_gradient(const double *ingrad) {
	update();
	const int inv = has_pinv ? 1 : 0, n = cat_count - inv;
	std::vector<double> q(n), dq(n);
	double sum = 0.0, dsum = 0.0;
	for (int i = 0; i < n; i++) {
		const double p = (2.0 * i + 1.0) / (2.0 * n);
		q[i] = quantile(p, shape);
		if (dist == RateDistribution::Gamma) {  // central difference in the shape, step cbrt(eps) * shape (sitemodel.c:266-289)
			const double eps = std::cbrt(std::numeric_limits<double>::epsilon());
			const double xp = shape * (1.0 + eps), xm = shape * (1.0 - eps);
			dq[i] = (quantile(p, xp) - quantile(p, xm)) / (xp - xm);
		} else
			dq[i] = -q[i] * std::log(-std::log(1.0 - p)) / (shape * shape);  // sitemodel.c:386
		sum += q[i] * cat_props[i + inv];
		dsum += dq[i] * cat_props[i + inv];
	}
	double g = 0.0;
	for (int i = 0; i < n; i++) g += ingrad[i + inv] * (dq[i] / sum - q[i] * dsum / sum / sum) * cat_props[i + inv];
	return g;
}

double SiteModel::pinv_gradient(const double *ingrad) {
	update();
	if (dist == RateDistribution::Constant) return ingrad[0] + ingrad[1] / cat_props[1];  // +I only
	const int n = cat_count - 1;
	double sum = 0.0, g = 0.0;
	for (int i = 0; i < n; i++) {
		const double q = quantile((2.0 * i + 1.0) / (2.0 * n), shape);
		sum += q;
		g += ingrad[i + 1] * q;
	}
	return ingrad[0] + g / (sum * (1.0 - cat_props[0]));
}

}  // namespace phyamd
