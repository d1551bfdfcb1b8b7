// tests/packer_asan.cpp -- TEST ONLY: AddressSanitizer + UBSan harness for the library's host packer (vargeno_amd/csrc/vg_hostpack*): random FASTQ text with
// malformed pieces (lines fgets would split, short quality lines, CRLF, truncated tails), cut anywhere into exact-size heap chunks so that any overread trips the
// sanitizer; compiled and run by tests/test_host_packer.py (GPU AddressSanitizer is not available on this pool: sanitizers run on the CPU build).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <string>
#include <vector>
#include "vg_hostpack.h"
int main()
{
	std::mt19937_64 rng(7);
	const char *B = "ACGTacgtNX";
	for (int trial = 0; trial < 60; trial++) {
		std::string text;
		const int nrec = 200 + (int)(rng() % 3000);
		for (int i = 0; i < nrec; i++) {
			const size_t L = (rng() % 5 == 0) ? rng() % 600 : 150;
			std::string r(L, 'A'), q(L, 'I');
			for (auto &c : r) c = B[rng() % (rng() % 50 == 0 ? 10 : 4)];
			for (auto &c : q) c = (char)(35 + rng() % 40);
			if (trial % 7 == 3 && i == nrec / 2) r.assign(1500, 'A');               // a line fgets would split
			if (trial % 11 == 5 && i == nrec / 3) q.assign(1, 'I');                  // a short quality line
			text += "@r" + std::to_string(i) + (trial % 3 == 1 ? "\r\n" : "\n") + r + (trial % 3 == 1 ? "\r\n" : "\n") + "+\n" + q + "\n";
		}
		if (trial % 5 == 2) text.resize(text.size() - 1 - rng() % 200);              // truncated tail
		for (int threads : {1, 3, 7}) {
			vgp::Packer pk(threads);
			pk.begin();
			size_t pos = 0;
			while (pos < text.size()) {
				size_t step = 1 + rng() % (rng() % 4 == 0 ? 50 : 400000);
				if (step > text.size() - pos) step = text.size() - pos;
				// exact-size heap copy so that any overread trips the sanitizer
				std::vector<uint8_t> chunk(text.begin() + (long)pos, text.begin() + (long)(pos + step));
				std::vector<uint64_t> km(vgp::Packer::kmers_cap(step)), me(vgp::Packer::reads_cap(step) + 1), of(vgp::Packer::reads_cap(step) + 2);
				vgp::Staging st; st.kmers = km.data(); st.kmers_cap = km.size(); st.meta = me.data(); st.offsets = of.data(); st.reads_cap = me.size();
				const vgp::ChunkResult r = pk.push(chunk.data(), step, st);
				if (r.n_reads && of[r.n_reads] != 32 * r.n_chunks) { printf("offsets mismatch\n"); return 1; }
				pos += step;
			}
			if (pk.consumed() > text.size()) { printf("consumed beyond the text\n"); return 1; }
		}
	}
	printf("packer sanitizer harness: ok\n");
	return 0;
}
