// ref_vote_replay.cc -- TEST INFRASTRUCTURE (never linked into the product).  The reference's OWN vote state machine
// (IndexTable + improved_index_table_add, src/qv.cc:57-178) driven with sequences of (index, kmer_pos, is_neighbor) read from a
// file, one fresh table and map per sequence; writes, per sequence, what the read loop looks at afterwards (qv.cc:1375-1380):
// whether a best entry exists, its index and (uint8_t) frequency, and the ambiguity flag.  Built by oracle/Makefile
// (`make vote_table`) from the reference's sources where they lie: this file only #includes src/qv.cc (its main() renamed).
//   input : u32 n_sequences, then per sequence u32 n, n x {u32 index, u32 kmer_pos, u32 is_neighbor}
//   output: per sequence {u32 has_best, u32 best_index, u32 best_freq, u32 ambiguous}
#define main vargeno_reference_main
#include "src/qv.cc"
#undef main

#include <stdio.h>
#include <stdint.h>

int main(int argc, char **argv)
{
	if (argc != 3) { fprintf(stderr, "usage: ref_vote_replay <in.bin> <out.bin>\n"); return 1; }
	FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
	if (!in || !out) return 1;
	uint32_t nseq = 0;
	if (fread(&nseq, 4, 1, in) != 1) return 1;
	IndexTable *table = (IndexTable *)malloc(sizeof(IndexTable));
	for (uint32_t s = 0; s < nseq; s++) {
		uint32_t n = 0;
		if (fread(&n, 4, 1, in) != 1) return 1;
		index_table_clear(table);
		unordered_map<uint32_t, unordered_set<uint32_t>> index_2_kmer_pos_set;
		for (uint32_t i = 0; i < n; i++) {
			uint32_t op[3];
			if (fread(op, 4, 3, in) != 3) return 1;
			improved_index_table_add(table, op[0], op[1], index_2_kmer_pos_set, op[2] != 0);
		}
		const uint32_t res[4] = {table->best != NULL, table->best ? table->best->index : 0u, table->best ? (uint32_t)table->best->freq : 0u, (uint32_t)table->ambiguous};
		fwrite(res, 4, 4, out);
	}
	fclose(in); fclose(out);
	return 0;
}
