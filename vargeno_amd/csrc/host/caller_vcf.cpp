// caller_vcf.cpp -- Bayesian genotype caller and VCF re-emitter with the reference's behaviour
// (src/qv.cc:1573-1747, 1789-1848).  Double precision and libm, exactly as upstream: GQ is an
// (int) truncation of -10*ln(confidence), so the arithmetic is kept operation for operation.
#include <math.h>
#include <stdio.h>

#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "vg_host.h"

namespace vgh {

static const int MAX_COV = 63;          // src/vartype.h:27
static const double ERR_RATE = 0.01;    // src/vartype.h:13
static const double AVG_COV = 7.1;      // src/vartype.h:14

Call choose_best_genotype(int ref_cnt, int alt_cnt, uint8_t ref_freq_enc, uint8_t alt_freq_enc)
{
	struct G { double g0, g1, g2; };
	static G cache[MAX_COV + 1][MAX_COV + 1];
	static double poisson[2 * MAX_COV + 1];
	static bool init = false;
	if (!init) {
		for (int r = 0; r <= MAX_COV; r++)
			for (int a = 0; a <= MAX_COV; a++) {
				cache[r][a].g0 = pow(1.0 - ERR_RATE, r) * pow(ERR_RATE, a);
				cache[r][a].g1 = pow(0.5, r + a);
				cache[r][a].g2 = pow(ERR_RATE, r) * pow(1.0 - ERR_RATE, a);
			}
		const double M = exp(-AVG_COV);
		for (int i = 0; i <= 2 * MAX_COV; i++) poisson[i] = (M * pow(AVG_COV, i)) / exp(lgamma(i + 1.0));
		init = true;
	}
	if ((ref_cnt == 0 && alt_cnt == 0) || (ref_cnt == MAX_COV && alt_cnt == MAX_COV)) return Call{0, 0.0};
	const double g0 = cache[ref_cnt][alt_cnt].g0, g1 = cache[ref_cnt][alt_cnt].g1, g2 = cache[ref_cnt][alt_cnt].g2;
	const double p = ref_freq_enc / 255.0, q = alt_freq_enc / 255.0;
	const double p2 = p * p, q2 = q * q;
	const double p_g0 = p2 * g0, p_g1 = (1.0 - p2 - q2) * g1, p_g2 = q2 * g2;
	const double total = p_g0 + p_g1 + p_g2;
	const int n = ref_cnt + alt_cnt;
	if (p_g0 > p_g1 && p_g0 > p_g2) return Call{1, ((double)(p_g0 / total)) * poisson[n]};
	if (p_g1 > p_g0 && p_g1 > p_g2) return Call{3, ((double)(p_g1 / total)) * poisson[n]};
	return Call{2, ((double)(p_g2 / total)) * poisson[n]};
}

// src/qv.cc:481-499: name = leading non-space characters (at most 32), length = atol of the rest
std::vector<ChrLen> read_chrlens(const std::string &path)
{
	std::vector<ChrLen> out;
	FILE *f = fopen(path.c_str(), "r");
	if (!f) throw Error{"cannot open " + path};
	char buf[256];
	while (fgets(buf, sizeof buf, f)) {
		size_t i = 0;
		std::string name;
		while (buf[i] && !isspace((unsigned char)buf[i]) && i < 32) name.push_back(buf[i++]);
		while (isspace((unsigned char)buf[i])) ++i;
		out.push_back(ChrLen{name, (uint64_t)atol(&buf[i])});
	}
	fclose(f);
	return out;
}

static std::vector<std::string> split(const std::string &text, char sep)     // src/allsome_util.cc:22-31
{
	std::vector<std::string> tokens;
	size_t start = 0, end = 0;
	while ((end = text.find(sep, start)) != std::string::npos) { tokens.push_back(text.substr(start, end - start)); start = end + 1; }
	tokens.push_back(text.substr(start));
	return tokens;
}

CallSummary write_genotyped_vcf(const SiteCounts &s, const std::vector<ChrLen> &chrlens, const std::string &vcf_in, const std::string &vcf_out)
{
	CallSummary sum;
	std::unordered_map<std::string, std::pair<char, double>> snp_2_genotype;
	for (size_t i = 0; i < s.pos.size(); i++) {                        // qv.cc:1573-1626
		uint64_t index = s.pos[i];
		size_t j;
		for (j = 0; j < chrlens.size() && index > chrlens[j].len; j++) index -= chrlens[j].len;
		const Call c = choose_best_genotype(s.ref_cnt[i], s.alt_cnt[i], s.ref_freq[i], s.alt_freq[i]);
		if (c.genotype == 0) continue;
		const std::string key = (j < chrlens.size() ? chrlens[j].name : std::string()) + "$" + std::to_string(index);
		char g = '0';
		if (c.genotype == 1) { ++sum.ref; g = '0'; } else if (c.genotype == 2) { ++sum.alt; g = '2'; } else { ++sum.het; g = '1'; }
		snp_2_genotype[key] = std::make_pair(g, c.confidence);
	}
	std::ifstream input(vcf_in);
	if (!input.good()) { fprintf(stderr, "Error opening: %s . You have failed.\n", vcf_in.c_str()); return sum; }
	std::ofstream output(vcf_out);
	std::string line;
	bool has_gt = false, has_gq = false, head_has_gt_col = true;
	int gt_index = -1, gq_index = -1;
	while (std::getline(input, line)) {                                  // qv.cc:1642-1745
		if (line.empty()) continue;
		if (line[0] == '#' && line[1] == '#') {
			output << line << "\n";
			if (line.find("ID=GT,") != std::string::npos) has_gt = true;
			else if (line.find("ID=GQ,") != std::string::npos) has_gq = true;
			continue;
		} else if (line[0] == '#') {
			if (!has_gt) { output << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">" << "\n"; gt_index = 0; }
			if (!has_gq) { output << "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">" << "\n"; gq_index = 1; }
			if (split(line, '\t').size() < 10) { head_has_gt_col = false; line += "\tFORMAT\tDONOR"; }
			output << line << "\n";
			continue;
		}
		std::vector<std::string> columns = split(line, '\t');
		std::string chr_name = columns[0];
		if (chr_name[0] != 'c') chr_name = "chr" + chr_name;
		if (columns.size() < 2) continue;
		const std::string key = chr_name + "$" + columns[1];
		auto it = snp_2_genotype.find(key);
		if (it == snp_2_genotype.end()) continue;
		std::string genotype_string = "0/0";
		if (it->second.first == '1') genotype_string = "0/1";
		else if (it->second.first == '2') genotype_string = "1/1";
		const int genotype_quality = -1 * 10 * log(it->second.second);   // qv.cc:1683, implicit double -> int
		std::vector<std::string> format_columns, info_columns;
		if (head_has_gt_col && columns.size() >= 10) { format_columns = split(columns[8], ':'); info_columns = split(columns[9], ':'); }
		if (gt_index == -1 && has_gt) {
			for (size_t i = 0; i < format_columns.size(); i++) if (format_columns[i] == "GT") { gt_index = (int)i; break; }
		}
		if (gt_index == -1 && has_gq) {                                  // (sic) the reference tests gt_index here too
			for (size_t i = 0; i < format_columns.size(); i++) if (format_columns[i] == "GQ") { gq_index = (int)i; break; }
		}
		if (has_gt && gt_index >= 0 && (size_t)gt_index < info_columns.size()) info_columns[(size_t)gt_index] = genotype_string;
		else if (!has_gt) { format_columns.push_back("GT"); info_columns.push_back(genotype_string); }
		if (has_gq && gq_index >= 0 && (size_t)gq_index < info_columns.size()) info_columns[(size_t)gq_index] = std::to_string(genotype_quality);
		else if (!has_gq) { format_columns.push_back("GQ"); info_columns.push_back(std::to_string(genotype_quality)); }
		std::string new_format = format_columns.empty() ? std::string() : format_columns[0];
		for (size_t i = 1; i < format_columns.size(); i++) new_format += ":" + format_columns[i];
		std::string new_info = info_columns.empty() ? std::string() : info_columns[0];
		for (size_t i = 1; i < info_columns.size(); i++) new_info += ":" + info_columns[i];
		if (head_has_gt_col && columns.size() >= 10) { columns[8] = new_format; columns[9] = new_info; }
		else { columns.push_back(new_format); columns.push_back(new_info); }
		std::string new_line = columns[0];
		for (size_t i = 1; i < columns.size(); i++) new_line += '\t' + columns[i];
		output << new_line << "\n";
	}
	return sum;
}

}  // namespace vgh
