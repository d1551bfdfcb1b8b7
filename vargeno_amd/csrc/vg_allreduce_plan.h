// vg_allreduce_plan.h -- the order of operations of vg_counts_allreduce_devices (one process, n index replicas, one exchange:
// the sum of the per-site counters), separated from the calls that carry it out so that it can be run against a
// mock on a machine without eight GPUs (tests/allreduce_mock.cpp; the product instantiates it with HIP + RCCL).
//
// RCCL wants one rank per device.  Replicas that share a device (small indexes, one-GPU test boxes) are therefore summed on
// that device first; the first replica of every device takes part in the collective; its guests copy the result afterwards.
//
// Backend (all calls return 0 on success):
//   int  device_of(int replica)
//   int  add_into(int dst_replica, int src_replica)          dst += src, both on dst's device, ordered on dst's stream
//   int  comm_init(const int *devices, int n_ranks)          one communicator per listed device
//   int  group_start(), group_end()
//   int  all_reduce(int replica, int rank)                   in-place sum over the communicator, on the replica's stream
//   int  copy_from(int dst_replica, int src_replica)         dst = src (same device), ordered on src's stream
//   int  sync(int replica)
//   void comm_destroy()
#pragma once
#include <vector>

namespace vg {

struct AllreducePlan {
	std::vector<int> rep_of;      // replica -> index into `reps`
	std::vector<int> reps;        // the first replica on every device, in order of appearance
};

inline AllreducePlan plan_allreduce(const std::vector<int> &device_of)
{
	AllreducePlan p;
	p.rep_of.assign(device_of.size(), -1);
	for (size_t i = 0; i < device_of.size(); i++) {
		for (size_t k = 0; k < p.reps.size(); k++) if (device_of[(size_t)p.reps[k]] == device_of[i]) p.rep_of[i] = (int)k;
		if (p.rep_of[i] < 0) { p.rep_of[i] = (int)p.reps.size(); p.reps.push_back((int)i); }
	}
	return p;
}

// returns 0, or the first failing step's code with *where naming it
template <class Backend>
int run_allreduce(Backend &B, int n, const char **where)
{
	std::vector<int> dev((size_t)n);
	for (int i = 0; i < n; i++) dev[(size_t)i] = B.device_of(i);
	const AllreducePlan p = plan_allreduce(dev);
	int rc = 0;
	*where = "";
	for (int i = 0; i < n && !rc; i++) {                      // every replica is idle: fold the guests into their device's first
		const int rep = p.reps[(size_t)p.rep_of[(size_t)i]];
		if (rep != i && (rc = B.add_into(rep, i))) *where = "sum of the replicas that share a device";
	}
	if (rc) return rc;
	const int nr = (int)p.reps.size();
	std::vector<int> devs((size_t)nr);
	for (int k = 0; k < nr; k++) devs[(size_t)k] = dev[(size_t)p.reps[(size_t)k]];
	if ((rc = B.comm_init(devs.data(), nr))) { *where = "communicator over the replicas' devices"; return rc; }
	rc = B.group_start();
	for (int k = 0; k < nr && !rc; k++) rc = B.all_reduce(p.reps[(size_t)k], k);
	const int rc2 = B.group_end();
	if (rc || rc2) { *where = "all-reduce of the site counters"; rc = rc ? rc : rc2; }
	for (int i = 0; i < n && !rc; i++) {                      // the guests take their device's result (same stream: after the all-reduce)
		const int rep = p.reps[(size_t)p.rep_of[(size_t)i]];
		if (rep != i && (rc = B.copy_from(i, rep))) *where = "copy of the reduced counters to a replica on the same device";
	}
	for (int k = 0; k < nr; k++) { const int s = B.sync(p.reps[(size_t)k]); if (s && !rc) { rc = s; *where = "stream synchronisation after the all-reduce"; } }
	B.comm_destroy();
	return rc;
}

}  // namespace vg
