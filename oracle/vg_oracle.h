/* vg_oracle.h -- CPU restatement of VarGeno's `geno` read loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (vargeno_amd/, libvargeno_hip.so, the vargeno CLI) never does.
 * Parity status: PINNED -- checked against the reference binary (oracle/_ref/vargeno, built by
 * oracle/Makefile from /root/reference) on the F-small fixture: tests/golden/fsmall.out.vcf.gz.
 */
#ifndef VG_ORACLE_H
#define VG_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vgo_index vgo_index;

/* Event counters; the unit costs of SURVEY.md §8(d) are applied to these to get algorithmic bytes. */
typedef struct {
	uint64_t reads;          /* FASTQ records submitted                                         */
	uint64_t reads_n;        /* skipped: N in the trimmed read (qv.cc:815-828)                   */
	uint64_t reads_invalid;  /* a non-ACGTN base (the reference aborts: util.c:103)              */
	uint64_t passes;         /* forward + reverse-complement passes run                          */
	uint64_t passes_ok;      /* passes that voted a unique position (qv.cc:1375)                 */
	uint64_t chunks;         /* 32-base chunks looked up                                         */
	uint64_t gate_open;      /* chunks with qual[c] < '8' (qv.cc:943)                            */
	uint64_t refbf_pos, snpbf_pos;   /* bit-vector probes that returned 1                         */
	uint64_t large_block;    /* gate-open chunks with ref bucket >= 100                          */
	uint64_t ref_query, snp_query;   /* query_ref_dict / query_snp_dict calls                     */
	uint64_t ref_probe, snp_probe;   /* bsearch element probes inside them                        */
	uint64_t scan_ref, scan_snp;     /* entries tested by the strided scans (B1)                   */
	uint64_t scan_oob;       /* of those, reads past the end of the array (treated as zero)      */
	uint64_t aux_ref, aux_snp;       /* aux rows fetched                                          */
	uint64_t site_test;      /* SNP-site tests on neighbour hits                                 */
	uint64_t ctx;            /* contexts appended                                                */
	uint64_t walks;          /* supporting contexts walked over the pileup (32 bases each)       */
	uint64_t incr;           /* counter increments (saturated ones included)                     */
	uint64_t ingest_bytes;   /* sum over reads of len/4 + ceil(len/32), len = trimmed length      */
} vgo_stats;

/* Build from SoA arrays (copied).  Layout = the reference's dict files, field by field
 * (dictgen.c:63-154, 156-275; reader qv.cc:519-695).  bf words = sdsl bit_vector payload. */
vgo_index *vgo_index_from_arrays(
	uint64_t n_ref, const uint64_t *ref_kmer, const uint32_t *ref_pos, const uint8_t *ref_amb,
	uint64_t n_ref_aux, const uint32_t *ref_aux /* [n_ref_aux][10] */,
	uint64_t n_snp, const uint64_t *snp_kmer, const uint32_t *snp_pos, const uint8_t *snp_info,
	const uint8_t *snp_amb, const uint8_t *snp_rf, const uint8_t *snp_af,
	uint64_t n_snp_aux, const uint32_t *snp_aux_pos /* [n][10] */, const uint8_t *snp_aux_info /* [n][10] */,
	uint64_t ref_bf_bits, const uint64_t *ref_bf_words,
	uint64_t snp_bf_bits, const uint64_t *snp_bf_words);

/* Load <prefix>.ref.dict / .snp.dict / .ref.bf / .snp.bf in the reference's on-disk format. */
vgo_index *vgo_index_load(const char *prefix);
void vgo_index_free(vgo_index *ix);

/* e = 1 turns the B1 strided scan into a plain bucket scan (to prove fixtures discriminate). */
void vgo_set_scan_stride(vgo_index *ix, int ref_stride, int snp_stride);

void vgo_reset_counts(vgo_index *ix);

/* Process n reads: bases/quals flat ASCII, offsets[n+1].  nthreads<=1 -> scalar loop.
 * Returns 0, or the number of reads with an invalid base (those are skipped). */
int64_t vgo_process(vgo_index *ix, const uint8_t *bases, const uint8_t *quals,
                    const uint64_t *offsets, uint64_t n_reads, int nthreads, vgo_stats *stats /* += */);

/* SNP sites (positions with ref != alt after seeding, ascending). */
uint64_t vgo_num_sites(const vgo_index *ix);
void vgo_get_sites(const vgo_index *ix, uint32_t *pos, uint8_t *ref_base, uint8_t *alt_base,
                   uint8_t *ref_freq, uint8_t *alt_freq, uint8_t *ref_cnt, uint8_t *alt_cnt);

/* Caller (qv.cc:1789-1848): returns 0 none, 1 = 0/0, 2 = 1/1, 3 = 0/1 (GTYPE_* order); *gq = (int)(-10 ln conf). */
int vgo_call(int ref_cnt, int alt_cnt, uint8_t ref_freq, uint8_t alt_freq, double *conf, int *gq);

/* Per-read trace for debugging parity: fills out[0]=passes run, out[1]=processed(0/1), out[2]=target index,
 * out[3]=#ref ctx, out[4]=#snp ctx of the LAST pass. */
void vgo_trace_read(vgo_index *ix, const uint8_t *bases, const uint8_t *quals, uint64_t len, uint32_t out[5]);

/* test hook: the vote state machine alone (qv.cc:132-178) on a sequence of adds; out = {has_best, best index, best freq, ambiguous} */
void vgo_vote_replay(const uint32_t *index, const uint32_t *kpos, const uint32_t *neigh, uint64_t n, uint32_t out[4]);

uint64_t vgo_alg_bytes(const vgo_stats *s);

#ifdef __cplusplus
}
#endif
#endif
