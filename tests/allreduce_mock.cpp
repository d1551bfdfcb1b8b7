// tests/allreduce_mock.cpp -- TEST ONLY (compiled and run by tests/test_distributed_cpu.py; never part of the product).
// Runs the product's own order of operations for the one multi-GPU exchange of the path (vargeno_amd/csrc/vg_allreduce_plan.h,
// which vg_counts_allreduce_devices instantiates with HIP + RCCL) against a mock of eight devices: "device memory" is host
// arrays, ncclAllReduce sums the participating replicas' arrays when the group ends -- and checks what RCCL would insist on:
// one rank per device, every rank of the communicator taking part exactly once, calls inside a group.
//   usage: allreduce_mock <device id of replica 0> <device id of replica 1> ...      prints "ok <ranks>" or a complaint
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <set>
#include <vector>

#include "../vargeno_amd/csrc/vg_allreduce_plan.h"

struct Mock {
	std::vector<int> dev;
	std::vector<std::vector<unsigned>> mem;     // one counter array per replica
	std::vector<int> comm_devs;
	std::vector<int> joined;                    // replica that joined as rank k (-1: none yet)
	bool in_group = false;
	int errors = 0;
	void complain(const char *what) { fprintf(stderr, "mock: %s\n", what); errors++; }
	int device_of(int i) { return dev[(size_t)i]; }
	int add_into(int dst, int src)
	{
		if (dev[(size_t)dst] != dev[(size_t)src]) complain("add_into across devices");
		for (size_t w = 0; w < mem[(size_t)dst].size(); w++) mem[(size_t)dst][w] += mem[(size_t)src][w];
		return 0;
	}
	int comm_init(const int *d, int n)
	{
		comm_devs.assign(d, d + n);
		if (std::set<int>(comm_devs.begin(), comm_devs.end()).size() != (size_t)n) complain("two ranks of the communicator on one device");
		joined.assign((size_t)n, -1);
		return 0;
	}
	int group_start() { if (in_group) complain("nested group"); in_group = true; return 0; }
	int all_reduce(int i, int rank)
	{
		if (!in_group) complain("all_reduce outside a group (n ranks in one thread would deadlock)");
		if (rank < 0 || rank >= (int)joined.size()) { complain("rank out of range"); return 1; }
		if (joined[(size_t)rank] != -1) complain("a rank joined twice");
		if (dev[(size_t)i] != comm_devs[(size_t)rank]) complain("replica joins with a communicator of another device");
		joined[(size_t)rank] = i;
		return 0;
	}
	int group_end()
	{
		in_group = false;
		for (int r : joined) if (r < 0) { complain("a rank of the communicator never joined: the collective would hang"); return 1; }
		std::vector<unsigned> sum(mem[0].size(), 0u);
		for (int r : joined) for (size_t w = 0; w < sum.size(); w++) sum[w] += mem[(size_t)r][w];
		for (int r : joined) mem[(size_t)r] = sum;
		return 0;
	}
	int copy_from(int dst, int src)
	{
		if (dev[(size_t)dst] != dev[(size_t)src]) complain("copy_from across devices");
		mem[(size_t)dst] = mem[(size_t)src];
		return 0;
	}
	int sync(int) { return 0; }
	void comm_destroy() { comm_devs.clear(); }
};

int main(int argc, char **argv)
{
	Mock m;
	for (int i = 1; i < argc; i++) m.dev.push_back(atoi(argv[i]));
	const int n = (int)m.dev.size();
	if (!n) return 2;
	const size_t W = 1000;
	std::vector<unsigned> want(W, 0u);
	for (int i = 0; i < n; i++) {
		std::vector<unsigned> a(W);
		for (size_t w = 0; w < W; w++) { a[w] = (unsigned)((i + 1) * 2654435761u + w * 40503u) % 97u; want[w] += a[w]; }
		m.mem.push_back(a);
	}
	const char *where = "";
	const int rc = vg::run_allreduce(m, n, &where);
	if (rc || m.errors) { printf("failed rc %d at '%s', %d complaints\n", rc, where, m.errors); return 1; }
	for (int i = 0; i < n; i++) if (m.mem[(size_t)i] != want) { printf("replica %d does not hold the sum\n", i); return 1; }
	printf("ok %zu\n", std::set<int>(m.dev.begin(), m.dev.end()).size());
	return 0;
}
