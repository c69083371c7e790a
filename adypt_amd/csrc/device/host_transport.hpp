// Host-staged transport with RCCL's call signatures — a TEST HOOK (ADYPT_COMM_TRANSPORT=host), never the product path.
//
// A GPU box of the test pool has ONE card and RCCL refuses two ranks on one device, so the process-per-GPU path (adypt_comm_*:
// counts exchange, strides, grouped send / receive, stream ordering, all-reduce, barrier) could only ever run with world = 1 there.
// With this table in place of librccl's, the very same code of multi.hip runs with world = N processes that may share a device:
// a send is a device-to-host copy ENQUEUED ON THE CALLER'S STREAM followed by a host function that publishes the bytes in a POSIX
// shared-memory mailbox; a receive is a host function on the caller's stream that waits for the mailbox, then a host-to-device copy —
// asynchronous and stream-ordered like the real collectives.  Single node, no performance claim.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace adypt_host_transport {

struct Mailbox {                     // one per (sender, receiver) pair and per use: created by the sender, unlinked by the receiver
	std::atomic<uint64_t> ready;     // bytes published (0 = not yet)
	unsigned char data[1];
};

struct Staging { void *p = nullptr; size_t cap = 0; }; // pinned: a stream-ordered copy from / to pageable memory would be staged at CALL time
struct HostComm {
	std::string name;                // from the 128-byte id
	int rank = 0, nranks = 1;
	Staging send_buf[64], recv_buf[64], all_buf; // one per peer and direction: operations on a stream are ordered, so each is free again
	                                             // by the time the next operation of the same kind and peer touches it
	uint64_t seq_send[64] = {0}, seq_recv[64] = {0}; // per peer: how many messages so far (names the mailbox)
	uint64_t seq_all = 0;
	std::atomic<int> failed{0};      // a peer never arrived / a mailbox could not be made: every later call on this communicator returns an error
	double timeout_s = 60.0;         // how long a host function waits for a peer before it gives up (ADYPT_HOST_TRANSPORT_TIMEOUT)
};

inline double &default_timeout_s() { static double t = 60.0; return t; } // set from Tunables::host_transport_timeout_s when the transport is selected (multi.hip)

// waits for the sender to publish; false after timeout_s (the peer died between creating the mailbox and filling it)
inline bool wait_ready(const Mailbox *m, double timeout_s)
{
	const auto t0 = std::chrono::steady_clock::now();
	while(m->ready.load(std::memory_order_acquire) == 0)
	{
		if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
		std::this_thread::sleep_for(std::chrono::microseconds(100));
	}
	return true;
}

inline size_t dtype_size(ncclDataType_t t) { return t == ncclDouble || t == ncclInt64 || t == ncclUint64 ? 8 : t == ncclFloat || t == ncclInt32 || t == ncclUint32 ? 4 : t == ncclHalf ? 2 : 1; }
inline std::string box_name(const HostComm *c, const char *kind, int src, int dst, uint64_t seq)
{
	char b[256];
	snprintf(b, sizeof b, "/adypt_%s_%s_%d_%d_%llu", c->name.c_str(), kind, src, dst, (unsigned long long)seq);
	return b;
}
inline Mailbox *map_box(const std::string &name, size_t bytes, bool create, double timeout_s = 120.0)
{
	const auto t0 = std::chrono::steady_clock::now();
	for(;;)
	{
		// the reader's clock is checked on EVERY turn, also while the box exists but is not sized yet (a creator that died between
		// shm_open and ftruncate must not keep the reader here for ever)
		if(!create && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return nullptr;
		int fd = shm_open(name.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
		if(fd >= 0)
		{
			const size_t total = sizeof(Mailbox) + bytes;
			if(create && ftruncate(fd, (off_t)total) != 0) { close(fd); return nullptr; }
			if(!create)
			{	// the creator may not have sized it yet
				struct stat st;
				if(fstat(fd, &st) != 0 || (size_t)st.st_size < total) { close(fd); std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }
			}
			void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			close(fd);
			return p == MAP_FAILED ? nullptr : (Mailbox *)p;
		}
		if(create) return nullptr;
		std::this_thread::sleep_for(std::chrono::microseconds(200));
	}
}

// (host functions must not call the HIP API: buffers are (re)allocated here, on the calling thread, and freed by CommDestroy)
inline void *staging(Staging &b, size_t bytes, hipStream_t stream)
{
	if(b.cap < bytes || !b.p)
	{
		(void)hipStreamSynchronize(stream);
		if(b.p) (void)hipHostFree(b.p);
		b.p = nullptr; b.cap = 0;
		if(hipHostMalloc(&b.p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
		b.cap = bytes ? bytes : 1;
	}
	return b.p;
}

struct Op { HostComm *c; void *staging; size_t bytes; int peer; uint64_t seq; int kind; ncclRedOp_t red; ncclDataType_t dt; size_t count; }; // freed by its host function

inline void host_send(void *p)
{
	Op *o = (Op *)p;
	const std::string n = box_name(o->c, "p2p", o->c->rank, o->peer, o->seq);
	if(Mailbox *m = map_box(n, o->bytes, true))
	{
		memcpy(m->data, o->staging, o->bytes);
		m->ready.store(o->bytes ? o->bytes : 1, std::memory_order_release);
		munmap(m, sizeof(Mailbox) + o->bytes);
	}
	else { o->c->failed.store(1); fprintf(stderr, "adypt host transport: cannot create mailbox %s\n", n.c_str()); }
	delete o;
}
inline void host_recv(void *p)
{
	Op *o = (Op *)p;
	const std::string n = box_name(o->c, "p2p", o->peer, o->c->rank, o->seq);
	Mailbox *m = map_box(n, o->bytes, false, o->c->timeout_s);
	if(m && wait_ready(m, o->c->timeout_s)) memcpy(o->staging, m->data, o->bytes);
	else
	{	// never copy an unwritten staging buffer to the device as if it were data: zeros, and the communicator is marked failed
		memset(o->staging, 0, o->bytes);
		o->c->failed.store(1);
		fprintf(stderr, "adypt host transport: rank %d never delivered %s\n", o->peer, n.c_str());
	}
	if(m) { munmap(m, sizeof(Mailbox) + o->bytes); shm_unlink(n.c_str()); }
	delete o;
}
inline void host_allreduce(void *p)
{
	Op *o = (Op *)p;
	HostComm *c = o->c;
	// every rank publishes its operand for every other rank, then reads everyone's: n x (n - 1) tiny mailboxes
	for(int r = 0; r < c->nranks; ++r)
	{
		if(r == c->rank) continue;
		const std::string n = box_name(c, "all", c->rank, r, o->seq);
		if(Mailbox *m = map_box(n, o->bytes, true)) { memcpy(m->data, o->staging, o->bytes); m->ready.store(1, std::memory_order_release); munmap(m, sizeof(Mailbox) + o->bytes); }
		else c->failed.store(1);
	}
	std::vector<unsigned char> acc((unsigned char *)o->staging, (unsigned char *)o->staging + o->bytes);
	for(int r = 0; r < c->nranks; ++r)
	{
		if(r == c->rank) continue;
		const std::string n = box_name(c, "all", r, c->rank, o->seq);
		Mailbox *m = map_box(n, o->bytes, false, c->timeout_s);
		if(!m || !wait_ready(m, c->timeout_s))
		{
			c->failed.store(1);
			fprintf(stderr, "adypt host transport: all-reduce: rank %d never arrived\n", r);
			if(m) { munmap(m, sizeof(Mailbox) + o->bytes); shm_unlink(n.c_str()); }
			continue;
		}
		if(o->dt == ncclDouble)
			for(size_t i = 0; i < o->count; ++i)
			{
				double a, b;
				memcpy(&a, acc.data() + 8 * i, 8); memcpy(&b, m->data + 8 * i, 8);
				a = o->red == ncclMax ? (a > b ? a : b) : a + b;
				memcpy(acc.data() + 8 * i, &a, 8);
			}
		munmap(m, sizeof(Mailbox) + o->bytes);
		shm_unlink(n.c_str());
	}
	memcpy(o->staging, acc.data(), o->bytes);
	delete o;
}

inline ncclResult_t GetUniqueId(ncclUniqueId *id)
{
	memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
	std::random_device rd;
	snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "%08x%08x_%d", rd(), rd(), (int)getpid());
	return ncclSuccess;
}
inline ncclResult_t CommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
	if(nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
	HostComm *c = new HostComm();
	c->name.assign(id.internal, strnlen(id.internal, NCCL_UNIQUE_ID_BYTES));
	c->rank = rank; c->nranks = nranks;
	c->timeout_s = default_timeout_s();
	*comm = (ncclComm_t)c;
	return ncclSuccess;
}
inline ncclResult_t CommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; } // the one-process form has its own test hook (ADYPT_MULTI_SHARED_DEVICE)
inline ncclResult_t CommDestroy(ncclComm_t comm)
{
	HostComm *c = (HostComm *)comm;
	(void)hipDeviceSynchronize();
	for(int i = 0; i < 64; ++i) { if(c->send_buf[i].p) (void)hipHostFree(c->send_buf[i].p); if(c->recv_buf[i].p) (void)hipHostFree(c->recv_buf[i].p); }
	if(c->all_buf.p) (void)hipHostFree(c->all_buf.p);
	delete c;
	return ncclSuccess;
}
inline ncclResult_t Send(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
	HostComm *c = (HostComm *)comm;
	const size_t bytes = count * dtype_size(dt);
	if(peer < 0 || peer >= 64) return ncclInvalidArgument;
	if(c->failed.load()) return ncclSystemError; // an earlier operation lost its peer: the caller sees ADYPT_E_HIP from adypt_comm_*
	void *st = staging(c->send_buf[peer], bytes, stream);
	if(!st) return ncclSystemError;
	if(bytes && hipMemcpyAsync(st, buf, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
	Op *o = new Op{c, st, bytes, peer, c->seq_send[peer]++, 0, ncclSum, dt, count};
	return hipLaunchHostFunc(stream, host_send, o) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
inline ncclResult_t Recv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
	HostComm *c = (HostComm *)comm;
	const size_t bytes = count * dtype_size(dt);
	if(peer < 0 || peer >= 64) return ncclInvalidArgument;
	if(c->failed.load()) return ncclSystemError;
	void *st = staging(c->recv_buf[peer], bytes, stream);
	if(!st) return ncclSystemError;
	Op *o = new Op{c, st, bytes, peer, c->seq_recv[peer]++, 1, ncclSum, dt, count};
	if(hipLaunchHostFunc(stream, host_recv, o) != hipSuccess) return ncclUnhandledCudaError;
	if(bytes && hipMemcpyAsync(buf, st, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
	return ncclSuccess;
}
inline ncclResult_t AllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
	HostComm *c = (HostComm *)comm;
	if(dt != ncclDouble || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
	if(c->failed.load()) return ncclSystemError;
	const size_t bytes = count * 8;
	void *st = staging(c->all_buf, bytes, stream);
	if(!st) return ncclSystemError;
	if(hipMemcpyAsync(st, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
	Op *o = new Op{c, st, bytes, -1, c->seq_all++, 2, op, dt, count};
	if(hipLaunchHostFunc(stream, host_allreduce, o) != hipSuccess) return ncclUnhandledCudaError;
	if(hipMemcpyAsync(recv, st, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
	return ncclSuccess;
}
inline ncclResult_t GroupStart() { return ncclSuccess; }
inline ncclResult_t GroupEnd() { return ncclSuccess; }
inline const char *GetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "host transport error"; }

}  // namespace adypt_host_transport
