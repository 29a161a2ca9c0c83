// The FMD index to every GPU of a node with RCCL over xGMI, from C (north_star: "RCCL broadcast of the
// FMD-index over xGMI ... host code stays in C").  The reference has no multi-GPU mode to follow (gasal_set_device is commented out,
// /root/reference/src/fastmap.c:143, src/bwamem.c:1938); this is the one collective of the design, outside the data path.
//
// Two shapes:
//   * one process per GPU (bmh_index_broadcast_rccl): every rank calls it with its communicator; the root hands over its index, the
//     others receive a 128-byte header, allocate, and receive the four arrays in one grouped broadcast;
//   * one process, N devices (bmh_index_replicate_all; what BMH_DEVICES=N uses): ncclCommInitAll + one grouped broadcast per array
//     across all devices -- a ring/tree over the xGMI links instead of N-1 hipMemcpyPeer copies out of device 0 one after the other.
//
// RCCL is resolved at run time, not linked: a process may already hold an RCCL (a C host linked with -lrccl, or PyTorch's bundled
// copy) and a communicator is only valid inside the library instance that made it.  Resolution order: the symbols already in the
// process's global namespace, then $BMH_RCCL_LIB, then librccl.so.1 / librccl.so of the loader path.  The bmh_rccl_* helpers create
// and destroy communicators through the SAME instance, so a host that uses them never has to care.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "bmh_internal.h"

namespace {
struct rccl_api_t {
	ncclResult_t (*GetUniqueId)(ncclUniqueId *);
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*CommCount)(const ncclComm_t, int *);
	ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
	ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
	ncclResult_t (*GroupStart)();
	ncclResult_t (*GroupEnd)();
	const char *(*GetErrorString)(ncclResult_t);
	const char *where;
	bool ok;
};

rccl_api_t g_api = {};
std::once_flag g_once;

bool bind_all(void *h, const char *where)
{
	rccl_api_t a = {};
#define BIND(f) *(void **)&a.f = dlsym(h, "nccl" #f); if (!a.f) return false
	BIND(GetUniqueId); BIND(CommInitRank); BIND(CommInitAll); BIND(CommDestroy); BIND(CommCount); BIND(CommUserRank);
	BIND(Broadcast); BIND(AllReduce); BIND(GroupStart); BIND(GroupEnd); BIND(GetErrorString);
#undef BIND
	a.where = where; a.ok = true;
	g_api = a;
	return true;
}

const rccl_api_t *rccl()
{
	std::call_once(g_once, [] {
		if (bind_all(RTLD_DEFAULT, "the process's global namespace")) return;
		const char *env = getenv("BMH_RCCL_LIB");
		const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *nm : names) {
			if (!nm || !*nm) continue;
			void *h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
			if (h && bind_all(h, nm)) return;
		}
	});
	if (!g_api.ok) { bmh_set_error("RCCL not found (tried the global namespace, $BMH_RCCL_LIB, librccl.so.1, librccl.so): %s", dlerror() ? dlerror() : "no such library"); return nullptr; }
	return &g_api;
}

#define NCK(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) { bmh_set_error("%s: %s", #call, R->GetErrorString(r_)); return BMH_ENODEV; } } while (0)
#define HCK(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #call, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

// what a receiver needs before it can allocate
struct bcast_hdr_t {
	uint64_t magic, primary, L2[5], seq_len, n_sa, l_pac, n_words;
	int32_t sa_shift, has_pac;
	uint8_t pad[128 - 8 * 11 - 8];
};
static_assert(sizeof(bcast_hdr_t) == 128, "header is one 128-byte message");

struct arrays_t { size_t bytes[4]; };      // blocks, sa, sa_bits, pac (allocation sizes, as bmh_index_upload pads them)
arrays_t array_bytes(const fmd_dev_t &f, bool has_pac)
{
	arrays_t a;
	a.bytes[0] = ((size_t)((f.seq_len + 63) / 64) + 1) * 32;
	a.bytes[1] = (size_t)f.n_sa * 4;
	a.bytes[2] = (size_t)(f.n_sa / 32 + 1) * 4;
	a.bytes[3] = has_pac ? (size_t)(f.l_pac / 4 + 1) + 16 : 0;
	return a;
}

// one array in pieces of at most 1 GiB (a dense hg38 suffix array is 25 GB): every piece one ncclBroadcast on `st`
int bcast_bytes(const rccl_api_t *R, const void *send, void *recv, size_t n, int root, ncclComm_t comm, hipStream_t st)
{
	const size_t CH = (size_t)1 << 30;
	for (size_t o = 0; o < n; o += CH) {
		const size_t c = n - o < CH ? n - o : CH;
		NCK(R->Broadcast(send ? (const uint8_t *)send + o : nullptr, (uint8_t *)recv + o, c, ncclUint8, root, comm, st));
	}
	return BMH_OK;
}
}   // namespace

extern "C" const char *bmh_rccl_where(void)
{
	const rccl_api_t *R = rccl();
	return R ? R->where : nullptr;
}

extern "C" int bmh_rccl_unique_id(void *id128)
{
	const rccl_api_t *R = rccl();
	if (!R) return BMH_ENODEV;
	if (!id128) { bmh_set_error("bmh_rccl_unique_id: null argument"); return BMH_EINVAL; }
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
	NCK(R->GetUniqueId((ncclUniqueId *)id128));
	return BMH_OK;
}

extern "C" int bmh_rccl_comm_init_rank(void **comm, int nranks, const void *id128, int rank)
{
	const rccl_api_t *R = rccl();
	if (!R) return BMH_ENODEV;
	if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { bmh_set_error("bmh_rccl_comm_init_rank: bad argument"); return BMH_EINVAL; }
	ncclUniqueId id; memcpy(&id, id128, sizeof(id));
	ncclComm_t c = nullptr;
	NCK(R->CommInitRank(&c, nranks, id, rank));
	*comm = (void *)c;
	return BMH_OK;
}

extern "C" void bmh_rccl_comm_destroy(void *comm)
{
	const rccl_api_t *R = rccl();
	if (R && comm) (void)R->CommDestroy((ncclComm_t)comm);
}

// Every rank of `comm` calls this on its own device.  root: `src` = its index; with out == NULL it only sends (in place), with out
// != NULL it also receives a fresh copy like everybody else (RCCL copies send -> recv on the root).  Other ranks: src ignored, *out =
// their own index (owned: bmh_index_free).  Blocks travel in the handle's native layout: no conversion on arrival.
extern "C" int bmh_index_broadcast_rccl(void *comm_, int root, const bmh_index_t *src, bmh_index_t **out, void *stream_)
{
	const rccl_api_t *R = rccl();
	if (!R) return BMH_ENODEV;
	ncclComm_t comm = (ncclComm_t)comm_;
	hipStream_t st = (hipStream_t)stream_;
	if (!comm) { bmh_set_error("bmh_index_broadcast_rccl: null communicator"); return BMH_EINVAL; }
	int nranks = 0, rank = -1;
	NCK(R->CommCount(comm, &nranks)); NCK(R->CommUserRank(comm, &rank));
	if (root < 0 || root >= nranks) { bmh_set_error("bmh_index_broadcast_rccl: root %d of %d ranks", root, nranks); return BMH_EINVAL; }
	const bool is_root = rank == root;
	if (is_root && !src) { bmh_set_error("bmh_index_broadcast_rccl: the root needs an index"); return BMH_EINVAL; }
	if (!is_root && !out) { bmh_set_error("bmh_index_broadcast_rccl: a receiving rank needs `out`"); return BMH_EINVAL; }
	if (out) *out = nullptr;
	// ---- header
	bcast_hdr_t h; memset(&h, 0, sizeof(h));
	if (is_root) {
		const fmd_dev_t &f = src->dev;
		h.magic = 0x424d48494458ull; h.primary = f.primary; memcpy(h.L2, f.L2, sizeof(h.L2)); h.seq_len = f.seq_len; h.n_sa = f.n_sa; h.l_pac = f.l_pac;
		h.n_words = src->n_words; h.sa_shift = f.sa_shift; h.has_pac = f.pac != nullptr;
	}
	// (the header's device buffer stays until the ranks have agreed that all are ready: the readiness flag below reuses it, so that a rank
	// which got this far needs no further allocation to take part in that reduction)
	void *d_hdr = nullptr;
	{
		HCK(hipMalloc(&d_hdr, sizeof(h)));
		bool ok = hipMemcpyAsync(d_hdr, &h, sizeof(h), hipMemcpyHostToDevice, st) == hipSuccess;
		ncclResult_t nr = ncclSuccess;
		if (ok) { nr = R->Broadcast(d_hdr, d_hdr, sizeof(h), ncclUint8, root, comm, st); ok = nr == ncclSuccess; }
		ok = ok && hipMemcpyAsync(&h, d_hdr, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
		if (!ok) (void)hipFree(d_hdr);
		if (!ok) { bmh_set_error("bmh_index_broadcast_rccl: header: %s", nr != ncclSuccess ? R->GetErrorString(nr) : hipGetErrorString(hipGetLastError())); return BMH_ENODEV; }
	}
	if (h.magic != 0x424d48494458ull) { (void)hipFree(d_hdr); bmh_set_error("bmh_index_broadcast_rccl: bad header from rank %d", root); return BMH_EINVAL; }
	// ---- arrays
	fmd_dev_t f; memset(&f, 0, sizeof(f));
	f.primary = h.primary; memcpy(f.L2, h.L2, sizeof(h.L2)); f.seq_len = h.seq_len; f.n_sa = h.n_sa; f.sa_shift = h.sa_shift; f.l_pac = h.l_pac;
	const arrays_t ab = array_bytes(f, h.has_pac != 0);
	const bool recv = out != nullptr;
	void *d[4] = {nullptr, nullptr, nullptr, nullptr};
	int alloc_rc = BMH_OK;
	if (recv) {
		for (int k = 0; k < 4 && alloc_rc == BMH_OK; ++k)
			if (ab.bytes[k] && hipMalloc(&d[k], ab.bytes[k]) != hipSuccess) {
				bmh_set_error("bmh_index_broadcast_rccl: %zu bytes of device memory: %s", ab.bytes[k], hipGetErrorString(hipGetLastError()));
				d[k] = nullptr; alloc_rc = BMH_ENOMEM;
			}
		if (alloc_rc == BMH_OK && d[3] && hipMemsetAsync(d[3], 0, ab.bytes[3], st) != hipSuccess) {       // (the text is read in aligned words past its last byte)
			bmh_set_error("bmh_index_broadcast_rccl: %s", hipGetErrorString(hipGetLastError()));
			alloc_rc = BMH_ENODEV;
		}
	}
	// Every rank says whether it is ready BEFORE the grouped broadcast: a rank that could not allocate used to leave here alone and its peers
	// waited in the broadcast for good.  One int per rank, minimum over the communicator: all go on, or all leave with an error.
	// The flag lives in the header's buffer (allocated before anything above could fail) and is set by a memset -- no host buffer, no staging: 0x01010101
	// = ready, 0 = not; a rank whose memset fails still JOINS the reduction (contributing what the buffer holds) and reports the failure on its side.
	{
		int *d_ok = (int *)d_hdr, ok_all = 0;
		bool fine = hipMemsetAsync(d_ok, alloc_rc == BMH_OK ? 1 : 0, sizeof(int), st) == hipSuccess;
		const ncclResult_t ar = R->AllReduce(d_ok, d_ok, 1, ncclInt32, ncclMin, comm, st);
		fine = fine && ar == ncclSuccess && hipMemcpyAsync(&ok_all, d_ok, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
		(void)hipFree(d_hdr); d_hdr = nullptr;
		ok_all = ok_all == 0x01010101 ? 1 : 0;
		if (!fine || ok_all != 1) {
			for (int k = 0; k < 4; ++k) if (d[k]) (void)hipFree(d[k]);
			if (alloc_rc == BMH_OK) bmh_set_error("bmh_index_broadcast_rccl: %s", fine ? "another rank could not allocate its copy of the index" : "the ranks could not agree that all are ready");
			return alloc_rc != BMH_OK ? alloc_rc : BMH_ENODEV;
		}
	}
	const void *s[4] = {nullptr, nullptr, nullptr, nullptr};
	if (is_root) { s[0] = src->dev.blocks; s[1] = src->dev.sa; s[2] = src->dev.sa_bits; s[3] = src->dev.pac; }
	int rc = BMH_OK;
	{
		const ncclResult_t g0 = R->GroupStart();
		if (g0 != ncclSuccess) { bmh_set_error("ncclGroupStart: %s", R->GetErrorString(g0)); rc = BMH_ENODEV; }
		for (int k = 0; k < 4 && rc == BMH_OK; ++k) {
			if (!ab.bytes[k]) continue;
			// the root's own buffers are exactly as long as the receivers' (bmh_index_upload / bmh_index_build pad alike); a root that
			// wraps caller memory (bmh_index_from_device) owns native blocks of that length too
			const void *sendp = is_root ? s[k] : nullptr;
			void *recvp = recv ? d[k] : (void *)s[k];             // send-only root: in place
			rc = bcast_bytes(R, is_root ? sendp : recvp, recvp, k == 3 ? (size_t)(f.l_pac / 4 + 1) : ab.bytes[k], root, comm, st);
		}
		const ncclResult_t g1 = R->GroupEnd();
		if (rc == BMH_OK && g1 != ncclSuccess) { bmh_set_error("ncclGroupEnd: %s", R->GetErrorString(g1)); rc = BMH_ENODEV; }
	}
	if (rc == BMH_OK && hipStreamSynchronize(st) != hipSuccess) { bmh_set_error("bmh_index_broadcast_rccl: %s", hipGetErrorString(hipGetLastError())); rc = BMH_ENODEV; }
	if (rc != BMH_OK) { for (int k = 0; k < 4; ++k) if (d[k]) (void)hipFree(d[k]); return rc; }
	if (recv) {
		bmh_index *ix = (bmh_index *)calloc(1, sizeof(bmh_index));
		ix->dev = f; ix->owns = true; ix->n_words = h.n_words;
		ix->dev.blocks = (const uint4 *)d[0]; ix->dev.sa = (const uint32_t *)d[1]; ix->dev.sa_bits = (const uint32_t *)d[2]; ix->dev.pac = (const uint8_t *)d[3];
		if (!h.has_pac) ix->dev.l_pac = 0;
		*out = ix;
	}
	return BMH_OK;
}

// One process, n devices: out[k] = the index on devices[k] (out[k] = src itself where devices[k] == src_device; the others are owned
// copies, freed with bmh_index_free while their device is current).  Distinct devices go through ONE communicator set
// (ncclCommInitAll) and a grouped broadcast per array; duplicates of a device in the list share its copy.  Falls back to
// bmh_index_replicate (hipMemcpyPeer) when RCCL cannot be resolved or refuses the device set (BMH_REPLICATE=peer forces that).
extern "C" int bmh_index_replicate_all(const bmh_index_t *src, int src_device, const int *devices, int n, bmh_index_t **out, int *used_rccl)
{
	if (!src || !devices || !out || n < 1) { bmh_set_error("bmh_index_replicate_all: bad argument"); return BMH_EINVAL; }
	if (used_rccl) *used_rccl = 0;
	int prev = 0;
	HCK(hipGetDevice(&prev));
	// distinct devices, the source first
	std::vector<int> devs{src_device};
	for (int k = 0; k < n; ++k) { bool seen = false; for (int d : devs) seen = seen || d == devices[k]; if (!seen) devs.push_back(devices[k]); }
	std::vector<bmh_index_t *> per_dev(devs.size(), nullptr);
	per_dev[0] = (bmh_index_t *)src;
	const int nd = (int)devs.size();
	const char *force = getenv("BMH_REPLICATE");
	const rccl_api_t *R = (nd > 1 && !(force && !strcmp(force, "peer"))) ? rccl() : nullptr;
	bool done = nd == 1;
	if (!done && R) {
		std::vector<ncclComm_t> comms((size_t)nd, nullptr);
		if (R->CommInitAll(comms.data(), nd, devs.data()) == ncclSuccess) {
			const fmd_dev_t &f = src->dev;
			const arrays_t ab = array_bytes(f, f.pac != nullptr);
			std::vector<std::vector<void *>> d((size_t)nd, std::vector<void *>(4, nullptr));
			std::vector<hipStream_t> st((size_t)nd, nullptr);
			bool ok = true;
			for (int i = 0; i < nd && ok; ++i) {
				ok = hipSetDevice(devs[i]) == hipSuccess && hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) == hipSuccess;
				for (int k = 0; k < 4 && ok && i > 0; ++k) if (ab.bytes[k]) ok = hipMalloc(&d[i][k], ab.bytes[k]) == hipSuccess && (k != 3 || hipMemsetAsync(d[i][k], 0, ab.bytes[k], st[i]) == hipSuccess);
			}
			const void *s[4] = {f.blocks, f.sa, f.sa_bits, f.pac};
			int rc = ok ? BMH_OK : BMH_ENOMEM;
			if (ok) {
				(void)R->GroupStart();
				for (int k = 0; k < 4 && rc == BMH_OK; ++k) {
					if (!ab.bytes[k]) continue;
					const size_t nbytes = k == 3 ? (size_t)(f.l_pac / 4 + 1) : ab.bytes[k];
					for (int i = 0; i < nd && rc == BMH_OK; ++i) {
						(void)hipSetDevice(devs[i]);
						rc = bcast_bytes(R, i == 0 ? s[k] : d[i][k], i == 0 ? (void *)s[k] : d[i][k], nbytes, 0, comms[i], st[i]);
					}
				}
				const ncclResult_t g1 = R->GroupEnd();
				if (rc == BMH_OK && g1 != ncclSuccess) { bmh_set_error("ncclGroupEnd: %s", R->GetErrorString(g1)); rc = BMH_ENODEV; }
				for (int i = 0; i < nd; ++i) { (void)hipSetDevice(devs[i]); if (hipStreamSynchronize(st[i]) != hipSuccess && rc == BMH_OK) { bmh_set_error("bmh_index_replicate_all: %s", hipGetErrorString(hipGetLastError())); rc = BMH_ENODEV; } }
			}
			for (int i = 0; i < nd; ++i) { (void)hipSetDevice(devs[i]); if (st[i]) (void)hipStreamDestroy(st[i]); (void)R->CommDestroy(comms[i]); }
			if (rc == BMH_OK) {
				for (int i = 1; i < nd; ++i) {
					bmh_index *ix = (bmh_index *)calloc(1, sizeof(bmh_index));
					*ix = *src; ix->owns = true; ix->owns_sa = false; ix->owns_blocks = false;
					ix->dev.blocks = (const uint4 *)d[i][0]; ix->dev.sa = (const uint32_t *)d[i][1]; ix->dev.sa_bits = (const uint32_t *)d[i][2]; ix->dev.pac = (const uint8_t *)d[i][3];
					per_dev[i] = ix;
				}
				done = true;
				if (used_rccl) *used_rccl = 1;
			} else {
				for (int i = 1; i < nd; ++i) { (void)hipSetDevice(devs[i]); for (void *p : d[i]) if (p) (void)hipFree(p); }
			}
		}
	}
	if (!done) {                                                   // no RCCL (or it refused): device-to-device copies, one after the other
		for (int i = 1; i < nd; ++i)
			if (bmh_index_replicate(src, src_device, devs[i], &per_dev[i]) != BMH_OK) {
				for (int q = 1; q < i; ++q) { (void)hipSetDevice(devs[q]); bmh_index_free(per_dev[q]); }
				(void)hipSetDevice(prev);
				return BMH_ENOMEM;
			}
	}
	(void)hipSetDevice(prev);
	for (int k = 0; k < n; ++k)
		for (int i = 0; i < nd; ++i) if (devs[i] == devices[k]) out[k] = per_dev[i];
	return BMH_OK;
}
