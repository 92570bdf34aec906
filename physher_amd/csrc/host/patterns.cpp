// patterns.cpp -- sequence encoding and site-pattern compression, bit-exact with the reference's order.
#include <cmath>
#include <cstring>

#include "phyamd_host.hpp"

namespace phyamd {

namespace {

// datatype.c:74-89: A C G T/U -> 0..3; R Y M W S K -> 5..10; B D H V N -> 11..15; other letters and '?' -> 16; rest -> 17
int nucleotide_code(unsigned char ch) {
	static const signed char letters[26] = {/*A*/ 0, /*B*/ 11, /*C*/ 1, /*D*/ 12, /*E*/ 16, /*F*/ 16, /*G*/ 2,  /*H*/ 13, /*I*/ 16,
	                                        /*J*/ 16, /*K*/ 10, /*L*/ 16, /*M*/ 7, /*N*/ 15, /*O*/ 16, /*P*/ 16, /*Q*/ 16, /*R*/ 5,
	                                        /*S*/ 9, /*T*/ 3, /*U*/ 3, /*V*/ 14, /*W*/ 8, /*X*/ 16, /*Y*/ 6, /*Z*/ 16};
	if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
	if (ch >= 'A' && ch <= 'Z') return letters[ch - 'A'];
	return ch == '?' ? 16 : 17;
}

// datatype.c:55-70 with the alphabet ACDEFGHIKLMNPQRSTVWY | B Z X * ? -
int amino_acid_code(unsigned char ch) {
	static const signed char letters[26] = {/*A*/ 0, /*B*/ 20, /*C*/ 1, /*D*/ 2, /*E*/ 3, /*F*/ 4, /*G*/ 5, /*H*/ 6, /*I*/ 7,
	                                        /*J*/ 24, /*K*/ 8, /*L*/ 9, /*M*/ 10, /*N*/ 11, /*O*/ 24, /*P*/ 12, /*Q*/ 13, /*R*/ 14,
	                                        /*S*/ 15, /*T*/ 16, /*U*/ 24, /*V*/ 17, /*W*/ 18, /*X*/ 22, /*Y*/ 19, /*Z*/ 21};
	if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
	if (ch >= 'A' && ch <= 'Z') return letters[ch - 'A'];
	if (ch == '*') return 23;
	if (ch == '?') return 24;
	return 25;
}

// growth sequence of the reference's table (hashtable.c:61-70)
const unsigned kPrimes[] = {5,        53,       97,        193,       389,       769,       1543,      3079,      6151,
                            12289,    24593,    49157,     98317,     196613,    393241,    786433,    1572869,   3145739,
                            6291469,  12582917, 25165843,  50331653,  100663319, 201326611, 402653189, 805306457, 1610612741};

unsigned mixed_hash(const uint8_t *v, int n) {
	unsigned h = v[0];  // sitepattern.c:71-79
	for (int i = 1; i < n; i++) h ^= v[i] + 0x9e3779b9u + (h << 6) + (h >> 2);
	h += ~(h << 9);  // hashtable.c:188-197
	h ^= (h >> 14) | (h << 18);
	h += h << 4;
	h ^= (h >> 10) | (h << 22);
	return h;
}

}  // namespace

int DataType::encode(const char *sym) const {
	switch (kind) {
		case DataTypeKind::Nucleotide: return nucleotide_code((unsigned char)sym[0]);
		case DataTypeKind::AminoAcid: return amino_acid_code((unsigned char)sym[0]);
		case DataTypeKind::Codon: {  // sitepattern.c:796-819, universal code: stops TAA TAG TGA are skipped
			const int a = nucleotide_code((unsigned char)sym[0]), b = nucleotide_code((unsigned char)sym[1]), c = nucleotide_code((unsigned char)sym[2]);
			if (a > 3 || b > 3 || c > 3) return 65;
			const int v = 16 * a + 4 * b + c;
			return v - (v > 48) - (v > 50) - (v > 56);
		}
		case DataTypeKind::General: {
			for (size_t i = 0; i < states.size(); i++)
				if (std::strncmp(states[i].c_str(), sym, (size_t)symbol_length) == 0 && (int)states[i].size() == symbol_length) return (int)i;
			return state_count;  // unknown
		}
	}
	return state_count;
}

int DataType::encode_string(const std::string &sym) const {
	if (kind != DataTypeKind::General) return encode(sym.c_str());
	for (size_t i = 0; i < states.size(); i++)
		if (states[i] == sym) return (int)i;
	for (size_t a = 0; a < ambiguities.size(); a++)
		if (ambiguities[a].first == sym) return state_count + (int)a;
	return state_count + (int)ambiguities.size();
}

void DataType::partial(int code, double *out) const {
	if (kind == DataTypeKind::Nucleotide) {  // datatype.h:26-66
		static const unsigned char mask[18] = {1, 2, 4, 8, 8, 5, 10, 3, 9, 6, 12, 14, 13, 11, 7, 15, 15, 15};
		const unsigned m = mask[code < 18 ? code : 17];
		for (int i = 0; i < 4; i++) out[i] = (m >> i) & 1u ? 1.0 : 0.0;
		return;
	}
	if (kind == DataTypeKind::General && code >= state_count && code - state_count < (int)ambiguities.size()) {  // _generic_partial
		for (int i = 0; i < state_count; i++) out[i] = 0.0;
		for (int st : ambiguities[code - state_count].second) out[st] = 1.0;
		return;
	}
	for (int i = 0; i < state_count; i++) out[i] = code >= state_count ? 1.0 : 0.0;
	if (code < state_count) out[code] = 1.0;
}

Patterns compress_patterns(const DataType &dt, const std::vector<std::string> &names, const std::vector<std::string> &sequences) {
	const int T = (int)sequences.size();
	if (T == 0 || names.size() != sequences.size()) throw Error("alignment needs one name per sequence");
	const size_t len = sequences[0].size();
	for (const auto &s : sequences)
		if (s.size() != len) throw Error("sequences are not aligned (different lengths)");
	const int step = dt.symbol_length;
	const int sites = (int)(len / step);
	// column-major codes: one contiguous key per site
	std::vector<uint8_t> cols((size_t)sites * T);
	for (int t = 0; t < T; t++)
		for (int s = 0; s < sites; s++) cols[(size_t)s * T + t] = (uint8_t)dt.encode(sequences[t].data() + (size_t)s * step);

	// The reference's chained table (hashtable.c): 193 buckets to start with (first prime >= 100), growth to the
	// next prime when a NEW key arrives at length == ceil(0.65 size); growth re-links every chain head-first (so
	// chains reverse); new keys go to the head of their chain; final order = buckets ascending, chains head->tail.
	struct Entry {
		int site;
		unsigned hash;
		int count;
		int next;
	};
	std::vector<Entry> entries;
	entries.reserve(sites);
	int prime = 3;
	unsigned size = kPrimes[prime];
	unsigned limit = (unsigned)std::ceil(size * 0.65);
	std::vector<int> head(size, -1);
	for (int s = 0; s < sites; s++) {
		const uint8_t *key = &cols[(size_t)s * T];
		const unsigned h = mixed_hash(key, T);
		int e = head[h % size];
		while (e >= 0 && !(entries[e].hash == h && std::memcmp(&cols[(size_t)entries[e].site * T], key, (size_t)T) == 0)) e = entries[e].next;
		if (e >= 0) {
			entries[e].count++;
			continue;
		}
		if (entries.size() == limit) {
			const unsigned nsize = kPrimes[++prime];
			std::vector<int> nhead(nsize, -1);
			for (unsigned b = 0; b < size; b++) {
				int x = head[b];
				while (x >= 0) {
					const int nx = entries[x].next;
					const unsigned nb = entries[x].hash % nsize;
					entries[x].next = nhead[nb];
					nhead[nb] = x;
					x = nx;
				}
			}
			head.swap(nhead);
			size = nsize;
			limit = (unsigned)std::ceil(size * 0.65);
		}
		entries.push_back(Entry{s, h, 1, head[h % size]});
		head[h % size] = (int)entries.size() - 1;
	}
	Patterns p;
	p.taxon_count = T;
	p.site_count = sites;
	p.pattern_count = (int)entries.size();
	p.names = names;
	p.states.resize((size_t)T * p.pattern_count);
	p.weights.resize(p.pattern_count);
	int k = 0;
	for (unsigned b = 0; b < size; b++)
		for (int e = head[b]; e >= 0; e = entries[e].next) {
			p.weights[k] = entries[e].count;
			for (int t = 0; t < T; t++) p.states[(size_t)t * p.pattern_count + k] = cols[(size_t)entries[e].site * T + t];
			k++;
		}
	return p;
}

}  // namespace phyamd
