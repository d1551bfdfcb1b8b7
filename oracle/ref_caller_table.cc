// ref_caller_table.cc -- TEST INFRASTRUCTURE (never linked into the product).  The reference's OWN genotype caller
// (choose_best_genotype, src/qv.cc:1789-1848, a static function) evaluated over its whole domain: every (ref_cnt, alt_cnt) in
// [0, 63]^2 for a list of encoded allele-frequency pairs, with the GQ the VCF pass derives from the confidence
// (src/qv.cc:1681).  Built by oracle/Makefile (`make caller_table`) from the reference's sources where they lie: this file only
// #includes src/qv.cc (its main() renamed) at compile time; nothing of it is copied.  Output: oracle/_ref/caller_table.bin,
// turned into tests/golden/caller_table.npz by tests/golden/make_caller_table.py.
//   record: u8 ref_freq, u8 alt_freq, u8 ref_cnt, u8 alt_cnt, i32 genotype, i32 gq, f64 confidence   (little-endian, packed)
#define main vargeno_reference_main
#include "src/qv.cc"
#undef main

#include <stdio.h>
#include <stdint.h>

int main(int argc, char **argv)
{
	if (argc != 2) { fprintf(stderr, "usage: ref_caller_table <out.bin>\n"); return 1; }
	FILE *f = fopen(argv[1], "wb");
	if (!f) return 1;
	static const int freqs[][2] = {{127, 127}, {0, 0}, {255, 255}, {255, 0}, {0, 255}, {242, 13}, {13, 242}, {229, 25}, {1, 254}, {254, 1},
	                               {200, 55}, {64, 191}, {178, 76}, {3, 3}, {128, 127}, {99, 155}};
	for (size_t k = 0; k < sizeof freqs / sizeof freqs[0]; k++)
		for (int rc = 0; rc <= 63; rc++)
			for (int ac = 0; ac <= 63; ac++) {
				const struct call c = choose_best_genotype(rc, ac, (uint8_t)freqs[k][0], (uint8_t)freqs[k][1]);
				const int gq = c.genotype == GTYPE_NONE ? 0 : (int)(-1 * 10 * log(c.confidence));      // qv.cc:1681
				const uint8_t h[4] = {(uint8_t)freqs[k][0], (uint8_t)freqs[k][1], (uint8_t)rc, (uint8_t)ac};
				const int32_t g = c.genotype, q = gq;
				const double conf = c.confidence;
				fwrite(h, 1, 4, f); fwrite(&g, 4, 1, f); fwrite(&q, 4, 1, f); fwrite(&conf, 8, 1, f);
			}
	fclose(f);
	return 0;
}
