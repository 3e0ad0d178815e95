// MultiGpu.cpp — one process, one host thread + one RCCL rank per GPU (include/aobaker_multi.h).
//
// The reference bakes on one device (BakeEngine.cpp:368 `IntersectionApi::Create(0)`); this is the north star's
// "texel tiles shard naturally across the 8 GPUs of one node with an RCCL gather over xGMI of the final AO image",
// as host C++ above the C ABI:
//   * every device owns an AobBake handle (scene uploaded to each, BVH built on each);
//   * per material each device rasterises + traces its own strip of texel rows (aob_bake_rasterize_rows + aob_bake_trace);
//   * ONE exchange step: the peers ncclSend their strip of AobResult cells, device 0 ncclRecvs them — grouped, straight
//     into its own result buffer at the strip's offset, each transfer on the peer's own xGMI link (16.8 MB per peer at
//     4096^2 / 8 GPUs) — then resolves, bleeds and downloads.  No ring, no all-gather: only device 0 needs the image.
// Ranks that share a device ordinal (a rehearsal of N ranks on one GPU: tests, `aobake_mgpu --oversubscribe`) cannot form
// an RCCL communicator (one rank per device); their strips move with hipMemcpyAsync instead — same orchestration, same cells.
#include "../../include/aobaker_multi.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>

namespace {

thread_local std::string t_error;
std::mutex g_error_mutex;
std::string g_error;

void set_error(const char* fmt, ...)
{
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	std::lock_guard<std::mutex> lock(g_error_mutex);
	g_error = buf;
}

double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A host thread bound to one device for the life of the bake: the main thread posts a job to every worker and waits.
class Worker
{
public:
	Worker() : thread([this] { loop(); }) {}
	~Worker()
	{
		{ std::lock_guard<std::mutex> lock(m); quit = true; }
		cv.notify_all();
		thread.join();
	}
	void post(std::function<void()> f)
	{
		{ std::lock_guard<std::mutex> lock(m); job = std::move(f); pending = true; done = false; }
		cv.notify_all();
	}
	void wait()
	{
		std::unique_lock<std::mutex> lock(m);
		cv.wait(lock, [this] { return done; });
	}
private:
	void loop()
	{
		for (;;)
		{
			std::function<void()> f;
			{
				std::unique_lock<std::mutex> lock(m);
				cv.wait(lock, [this] { return pending || quit; });
				if (quit && !pending) return;
				f = std::move(job); pending = false;
			}
			f();
			{ std::lock_guard<std::mutex> lock(m); done = true; }
			cv.notify_all();
		}
	}
	std::mutex m;
	std::condition_variable cv;
	std::function<void()> job;
	bool pending = false, done = true, quit = false;
	std::thread thread;
};

} // namespace

std::vector<int32_t> aob_equal_rows(int32_t height, int parts, int32_t align)
{
	const int32_t tile_rows = (height + align - 1) / align;
	std::vector<int32_t> b((size_t)parts + 1);
	for (int r = 0; r <= parts; r++)
		b[(size_t)r] = std::min<int64_t>((int64_t)tile_rows * r / parts * align, height);
	b[(size_t)parts] = height;
	return b;
}

std::vector<int32_t> aob_rebalance_rows(const std::vector<int32_t>& bounds, const std::vector<double>& old_ms, int32_t height, int32_t align)
{
	const size_t parts = old_ms.size();
	if (bounds.size() != parts + 1 || parts == 0) return bounds;
	double total = 0.0;
	for (size_t r = 0; r < parts; r++) total += std::max(0.0, old_ms[r]);
	if (!(total > 0.0)) return bounds;
	std::vector<int32_t> out(parts + 1);
	out[0] = 0; out[parts] = height;
	// cumulative cost over the rows: piecewise linear, slope old_ms[r] / rows(r) inside old strip r
	size_t seg = 0; double before = 0.0;            // cost of the old strips before `seg`
	for (size_t r = 1; r < parts; r++)
	{
		const double want = total * (double)r / (double)parts;
		while (seg + 1 < parts && before + std::max(0.0, old_ms[seg]) < want) { before += std::max(0.0, old_ms[seg]); seg++; }
		const double span = (double)(bounds[seg + 1] - bounds[seg]);
		const double cost = std::max(0.0, old_ms[seg]);
		double row = cost > 0.0 ? bounds[seg] + span * (want - before) / cost : (double)bounds[seg + 1];
		int32_t cut = (int32_t)((int64_t)(row / align + 0.5) * align);
		cut = std::max(cut, out[r - 1]);            // monotone; a strip may become empty
		cut = std::min(cut, height);
		out[r] = cut;
	}
	return out;
}

struct AobMultiGpuBake::Impl
{
	int n = 0;
	bool balance = true, by_rows = true, use_rccl = false;
	AobOptions opt;
	std::vector<int> devices;
	std::vector<AobMaterialDesc> materials;
	std::vector<Worker*> workers;
	std::vector<AobBake*> bakes;
	std::vector<hipStream_t> streams;
	std::vector<ncclComm_t> comms;
	std::vector<std::vector<uint8_t*>> results;          // [rank][material] device pointer of the AobResult buffer
	std::vector<std::vector<int32_t>> rows;              // [material] strip cuts, n + 1 entries
	std::vector<int> status;
	std::vector<std::string> errors;
	std::atomic<bool> giving_up{false};                  // a rank's exchange failed or timed out: the others stop waiting
	bool broken = false;                                 // the communicators were aborted: no further Step
	double exchange_timeout_ms = 30000.0;

	// After any exchange error the communicators are ABORTED: a rank must not sit in a grouped ncclRecv (or a stream that
	// holds one) for a send that will never come.  The bake is unusable afterwards and says so.
	void abort_comms()
	{
		for (ncclComm_t& c : comms) if (c) { ncclCommAbort(c); c = nullptr; }
		broken = true;
	}

	template <class F> int on_all(F f)
	{
		for (int r = 0; r < n; r++)
			workers[(size_t)r]->post([this, r, f] {
				status[(size_t)r] = f(r);
				if (status[(size_t)r] != AOB_OK && errors[(size_t)r].empty()) errors[(size_t)r] = aob_last_error();
			});
		for (int r = 0; r < n; r++) workers[(size_t)r]->wait();
		for (int r = 0; r < n; r++)
			if (status[(size_t)r] != AOB_OK)
			{
				set_error("device %d (rank %d): %s", devices[(size_t)r], r, errors[(size_t)r].c_str());
				errors[(size_t)r].clear();
				return status[(size_t)r];
			}
		return AOB_OK;
	}
};

AobMultiGpuBake::AobMultiGpuBake() : impl(new Impl()) {}

const char* AobMultiGpuBake::LastError()
{
	std::lock_guard<std::mutex> lock(g_error_mutex);
	t_error = g_error;
	return t_error.c_str();
}

int AobMultiGpuBake::DeviceCount() const { return impl->n; }

AobMultiGpuBake* AobMultiGpuBake::Create(const AobSceneDesc* scene, const AobOptions* options, const std::vector<int>& devices, bool balance)
{
	if (!scene || !options || devices.empty()) { set_error("scene, options or device list missing"); return nullptr; }
	AobMultiGpuBake* self = new AobMultiGpuBake();
	Impl& d = *self->impl;
	d.n = (int)devices.size();
	d.devices = devices;
	d.balance = balance;
	d.opt = *options;
	// REFERENCE_COMPAT numbers the hemisphere table by a surfel's position in its section's list: strips of rows would
	// renumber them, so that sampler shards the list instead (every device rasterises the whole atlas)
	d.by_rows = options->sampler != AOB_SAMPLER_REFERENCE_COMPAT;
	d.materials.assign(scene->materials, scene->materials + scene->material_count);
	d.bakes.assign((size_t)d.n, nullptr);
	d.streams.assign((size_t)d.n, nullptr);
	d.status.assign((size_t)d.n, AOB_OK);
	d.errors.assign((size_t)d.n, std::string());
	d.results.assign((size_t)d.n, std::vector<uint8_t*>(scene->material_count, nullptr));
	for (uint32_t i = 0; i < scene->material_count; i++) d.rows.push_back(aob_equal_rows(scene->materials[i].height, d.n));
	int visible = 0;
	if (hipGetDeviceCount(&visible) != hipSuccess) visible = 0;
	for (int dev : devices)
		if (dev < 0 || dev >= visible) { set_error("device %d is not on this box (%d visible)", dev, visible); delete self; return nullptr; }
	const bool distinct = std::set<int>(devices.begin(), devices.end()).size() == devices.size();
	d.use_rccl = distinct && d.n > 1;
	if (!distinct) d.balance = false;      // ranks that share a device time each other's work: nothing to balance by
	if (d.use_rccl)
	{
		d.comms.assign((size_t)d.n, nullptr);
		const ncclResult_t rc = ncclCommInitAll(d.comms.data(), d.n, d.devices.data());
		if (rc != ncclSuccess) { set_error("ncclCommInitAll over %d devices failed: %s", d.n, ncclGetErrorString(rc)); d.comms.clear(); delete self; return nullptr; }
	}
	for (int r = 0; r < d.n; r++) d.workers.push_back(new Worker());
	const int rc = d.on_all([&d, scene](int r) -> int {
		AobDeviceConfig cfg; std::memset(&cfg, 0, sizeof cfg);
		cfg.device = d.devices[(size_t)r];
		int s = aob_init(&cfg);
		if (s != AOB_OK) return s;
		// only rank 0 downloads: the peers get no host pointers
		std::vector<AobMaterialDesc> mats(d.materials);
		if (r != 0) for (auto& m : mats) m.data = nullptr;
		AobSceneDesc local = *scene;
		local.materials = mats.data();
		s = aob_bake_create(&local, &d.opt, &d.bakes[(size_t)r]);
		if (s != AOB_OK) return s;
		void* stream = nullptr;
		s = aob_bake_stream(d.bakes[(size_t)r], &stream);
		if (s != AOB_OK) return s;
		d.streams[(size_t)r] = (hipStream_t)stream;
		for (uint32_t i = 0; i < scene->material_count; i++)
		{
			void* p = nullptr; uint64_t bytes = 0;
			s = aob_bake_results_device_ptr(d.bakes[(size_t)r], (int32_t)i, &p, &bytes);
			if (s != AOB_OK) return s;
			d.results[(size_t)r][i] = (uint8_t*)p;
		}
		return AOB_OK;
	});
	if (rc != AOB_OK) { delete self; return nullptr; }
	return self;
}

AobMultiGpuBake::~AobMultiGpuBake()
{
	Impl& d = *impl;
	if (!d.workers.empty())
		d.on_all([&d](int r) -> int {
			if (d.bakes[(size_t)r]) { aob_bake_destroy(d.bakes[(size_t)r]); d.bakes[(size_t)r] = nullptr; }
			if (r < (int)d.comms.size() && d.comms[(size_t)r]) { ncclCommDestroy(d.comms[(size_t)r]); d.comms[(size_t)r] = nullptr; }
			aob_shutdown();
			return AOB_OK;
		});
	for (Worker* w : d.workers) delete w;
	delete impl;
}

int AobMultiGpuBake::Step(AobMultiStepStats* stats)
{
	Impl& d = *impl;
	if (d.broken) { set_error("an earlier exchange failed and the RCCL communicators were aborted: create the bake again"); return AOB_ERR_DEVICE; }
	const int n = d.n;
	const size_t nmat = d.materials.size();
	std::vector<double> rank_ms((size_t)n, 0.0), gather_ms((size_t)n, 0.0);
	std::vector<float> trace_ms((size_t)n, 0.0f);
	std::vector<uint64_t> rank_surfels((size_t)n, 0);
	std::vector<std::vector<double>> mat_ms(nmat, std::vector<double>((size_t)n, 0.0));
	const double t0 = now_ms();
	int rc = d.on_all([&](int r) -> int {
		const double a = now_ms();
		const int s = aob_bake_build_bvh(d.bakes[(size_t)r]);
		rank_ms[(size_t)r] += now_ms() - a;          // enqueue only; the build's device time lands in the first material's sync
		return s;
	});
	if (rc != AOB_OK) return rc;
	for (size_t i = 0; i < nmat; i++)
	{
		const int32_t W = d.materials[i].width, H = d.materials[i].height;
		const uint64_t cells = (uint64_t)W * H;
		// every rank's cell range [first, last] of this material (first > last: nothing to send)
		std::vector<uint64_t> first((size_t)n, 1), last((size_t)n, 0);
		std::vector<int32_t> overlapping((size_t)n, 0);
		// phase 1: rasterise + trace the rank's share, wait for it
		rc = d.on_all([&, i](int r) -> int {
			AobBake* b = d.bakes[(size_t)r];
			const double a = now_ms();
			int s;
			uint64_t count = 0;
			if (d.by_rows)
			{
				const int32_t r0 = d.rows[i][(size_t)r], r1 = d.rows[i][(size_t)r + 1];
				s = aob_bake_rasterize_rows(b, (int32_t)i, r0, r1);
				if (s == AOB_OK) s = aob_bake_trace(b, (int32_t)i, 0, UINT64_MAX);
				if (s == AOB_OK) s = aob_bake_surfel_count(b, (int32_t)i, &count);
				if (r1 > r0) { first[(size_t)r] = (uint64_t)r0 * W; last[(size_t)r] = (uint64_t)r1 * W - 1; }
			}
			else
			{
				s = aob_bake_rasterize(b, (int32_t)i);
				std::vector<uint64_t> bounds((size_t)n + 1), c0((size_t)n), c1((size_t)n);
				int32_t ov = 0;
				if (s == AOB_OK) s = aob_bake_shard_bounds(b, (int32_t)i, (uint32_t)n, bounds.data(), c0.data(), c1.data(), &ov);
				if (s == AOB_OK) s = aob_bake_trace(b, (int32_t)i, bounds[(size_t)r], bounds[(size_t)r + 1]);
				count = bounds[(size_t)r + 1] - bounds[(size_t)r];
				if (s == AOB_OK && r == 0)      // identical on every rank (same rasterisation); rank 0's copy is the one used
					for (int p = 0; p < n; p++) { first[(size_t)p] = c0[(size_t)p]; last[(size_t)p] = c1[(size_t)p]; overlapping[(size_t)p] = ov; }
			}
			if (s == AOB_OK) s = aob_bake_sync(b);
			if (s != AOB_OK) return s;
			const double ms = now_ms() - a;
			mat_ms[i][(size_t)r] = ms; rank_ms[(size_t)r] += ms;
			float t = 0.f;
			if (aob_bake_last_trace_ms(b, &t) == AOB_OK) trace_ms[(size_t)r] += t;
			rank_surfels[(size_t)r] += count;
			return AOB_OK;
		});
		if (rc != AOB_OK) return rc;
		const bool reduce = !d.by_rows && overlapping[0] != 0;
		// phase 2a: every rank ENQUEUES its side of the one exchange step (nothing waits here)
		d.giving_up = false;
		std::vector<double> posted((size_t)n, 0.0);          // when each rank began to post its side of the exchange (its gather time starts there)
		rc = d.on_all([&, i, reduce](int r) -> int {
			uint8_t* mine = d.results[(size_t)r][i];
			hipStream_t stream = d.streams[(size_t)r];
			posted[(size_t)r] = now_ms();
			if (n > 1 && d.use_rccl)
			{
				ncclResult_t e = ncclSuccess;
				if (reduce)
				{
					// sections of one material overlap and the list was cut by surfel count: cells of different ranks may
					// coincide — sum them (x + 0 is exact for the cells only one rank touched).  QUANTISED sums are integers:
					// the atlas is the single-device bake's, byte for byte.  FLOAT cells are floats rounded once from each
					// rank's exact sums, so a cell two ranks both touched is the sum of two rounded floats: equal to the
					// single-device cell within 1 ulp per extra rank (aobaker_multi.h states it; --check-single reports it).
					e = ncclReduce(mine, mine, cells * 2, d.opt.accumulate == AOB_ACCUMULATE_QUANTISED ? ncclInt32 : ncclFloat32, ncclSum, 0,
					               d.comms[(size_t)r], stream);
				}
				else if (r == 0)
				{
					e = ncclGroupStart();
					for (int p = 1; p < n && e == ncclSuccess; p++)
						if (first[(size_t)p] <= last[(size_t)p])
							e = ncclRecv(mine + first[(size_t)p] * sizeof(AobResult), (last[(size_t)p] - first[(size_t)p] + 1) * sizeof(AobResult),
							             ncclUint8, p, d.comms[0], stream);
					const ncclResult_t e2 = ncclGroupEnd();
					if (e == ncclSuccess) e = e2;
				}
				else if (first[(size_t)r] <= last[(size_t)r])
				{
					e = ncclSend(mine + first[(size_t)r] * sizeof(AobResult), (last[(size_t)r] - first[(size_t)r] + 1) * sizeof(AobResult),
					             ncclUint8, 0, d.comms[(size_t)r], stream);
				}
				if (e != ncclSuccess) { d.errors[(size_t)r] = std::string("RCCL: ") + ncclGetErrorString(e); return AOB_ERR_DEVICE; }
			}
			else if (n > 1 && r == 0)
			{
				// ranks sharing a device (rehearsal): the strips are already in this device's memory
				if (reduce)
				{
					// the sum-reduce on one device: the peers' whole cell arrays are added to rank 0's in rank order, on rank 0's
					// stream (every rank's trace was waited for in phase 1)
					// (aob_bake_add_results_from orders each sum behind the peer's own stream besides)
					for (int p = 1; p < n; p++)
						if (aob_bake_add_results_from(d.bakes[0], (int32_t)i, d.bakes[(size_t)p], (int32_t)i) != AOB_OK) { d.errors[0] = aob_last_error(); return AOB_ERR_DEVICE; }
				}
				else for (int p = 1; p < n; p++)
					if (first[(size_t)p] <= last[(size_t)p])
					{
						const hipError_t e = hipMemcpyAsync(mine + first[(size_t)p] * sizeof(AobResult), d.results[(size_t)p][i] + first[(size_t)p] * sizeof(AobResult),
						                                    (last[(size_t)p] - first[(size_t)p] + 1) * sizeof(AobResult), hipMemcpyDeviceToDevice, stream);
						if (e != hipSuccess) { d.errors[0] = std::string("hipMemcpyAsync: ") + hipGetErrorString(e); return AOB_ERR_DEVICE; }
					}
			}
			return AOB_OK;
		});
		// a rank that could not enqueue its send leaves rank 0's grouped receive without a partner: abort, never wait
		if (rc != AOB_OK) { if (d.use_rccl) d.abort_comms(); return rc; }
		// phase 2b: wait for the exchange by POLLING the stream and the communicator's asynchronous error (a blocking stream
		// sync cannot be called off), then device 0 resolves
		rc = d.on_all([&, i](int r) -> int {
			AobBake* b = d.bakes[(size_t)r];
			if (n > 1 && d.use_rccl)
			{
				const double t = now_ms();
				for (;;)
				{
					const hipError_t q = hipStreamQuery(d.streams[(size_t)r]);
					if (q == hipSuccess) break;
					if (q != hipErrorNotReady) { d.errors[(size_t)r] = std::string("hipStreamQuery: ") + hipGetErrorString(q); d.giving_up = true; return AOB_ERR_DEVICE; }
					ncclResult_t async = ncclSuccess;
					const ncclResult_t e = ncclCommGetAsyncError(d.comms[(size_t)r], &async);
					if (e != ncclSuccess || async != ncclSuccess)
					{
						d.errors[(size_t)r] = std::string("RCCL (asynchronous): ") + ncclGetErrorString(e != ncclSuccess ? e : async);
						d.giving_up = true;
						return AOB_ERR_DEVICE;
					}
					if (d.giving_up) { d.errors[(size_t)r] = "another rank's exchange failed"; return AOB_ERR_DEVICE; }
					if (now_ms() - t > d.exchange_timeout_ms)
					{
						d.errors[(size_t)r] = "the exchange of result strips did not finish within " + std::to_string((int)(d.exchange_timeout_ms / 1000.0)) + " s";
						d.giving_up = true;
						return AOB_ERR_DEVICE;
					}
					std::this_thread::sleep_for(std::chrono::microseconds(20));
				}
				gather_ms[(size_t)r] += now_ms() - posted[(size_t)r];
			}
			else if (n > 1 && r == 0)
			{
				// ranks sharing a device: the strips were copied (or summed) on device 0's stream — its gather ends when that stream has drained
				if (hipStreamSynchronize(d.streams[0]) != hipSuccess) { d.errors[0] = "hipStreamSynchronize after the strips' copies failed"; return AOB_ERR_DEVICE; }
				gather_ms[0] += now_ms() - posted[0];
			}
			int s = AOB_OK;
			if (r == 0) s = aob_bake_resolve(b, (int32_t)i);
			if (s == AOB_OK) s = aob_bake_sync(b);
			return s;
		});
		if (rc != AOB_OK) { if (d.use_rccl && d.giving_up) d.abort_comms(); return rc; }
	}
	const double wall = now_ms() - t0;
	if (stats)
	{
		stats->wall_ms = wall;
		stats->trace_ms = trace_ms; stats->rank_ms = rank_ms; stats->rank_surfels = rank_surfels;
		stats->rank_gather_ms = gather_ms;
		stats->rank_bvh_build_ms.assign((size_t)n, 0.f);
		for (int r = 0; r < n; r++) { AobStats s1; if (aob_bake_get_stats(d.bakes[(size_t)r], &s1) == AOB_OK) stats->rank_bvh_build_ms[(size_t)r] = s1.ms_bvh_build; }
		stats->surfels = 0;
		for (uint64_t c : rank_surfels) stats->surfels += c;
		stats->rays = stats->surfels * (uint64_t)d.opt.sample_count;
		stats->row_begin.clear(); stats->row_end.clear();
		if (nmat)
			for (int r = 0; r < n; r++) { stats->row_begin.push_back(d.rows[0][(size_t)r]); stats->row_end.push_back(d.rows[0][(size_t)r + 1]); }
	}
	if (d.balance && d.by_rows && n > 1)
		for (size_t i = 0; i < nmat; i++)
			d.rows[i] = aob_rebalance_rows(d.rows[i], mat_ms[i], d.materials[i].height);
	return AOB_OK;
}

int AobMultiGpuBake::Download()
{
	Impl& d = *impl;
	std::vector<int> st(1, AOB_OK);
	d.workers[0]->post([&d, &st] {
		for (size_t i = 0; i < d.materials.size() && st[0] == AOB_OK; i++)
			if (d.materials[i].data) st[0] = aob_bake_download(d.bakes[0], (int32_t)i, nullptr, 0);
		if (st[0] != AOB_OK) d.errors[0] = aob_last_error();
	});
	d.workers[0]->wait();
	if (st[0] != AOB_OK) set_error("device %d: %s", d.devices[0], d.errors[0].c_str());
	return st[0];
}

int AobMultiGpuBake::CopyResults(int32_t material, AobResult* dst, uint64_t capacity)
{
	Impl& d = *impl;
	int st = AOB_OK;
	d.workers[0]->post([&] {
		st = aob_bake_copy_results(d.bakes[0], material, dst, capacity);
		if (st != AOB_OK) d.errors[0] = aob_last_error();
	});
	d.workers[0]->wait();
	if (st != AOB_OK) set_error("device %d: %s", d.devices[0], d.errors[0].c_str());
	return st;
}

int AobMultiGpuBake::GetStats(AobStats* stats)
{
	Impl& d = *impl;
	if (!stats) { set_error("stats is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
	std::vector<AobStats> all((size_t)d.n);
	const int rc = d.on_all([&](int r) -> int { return aob_bake_get_stats(d.bakes[(size_t)r], &all[(size_t)r]); });
	if (rc != AOB_OK) return rc;
	*stats = all[0];
	if (d.by_rows)
	{
		stats->surfel_count = 0;
		for (const AobStats& s : all) stats->surfel_count += s.surfel_count;
		stats->ray_count = stats->surfel_count * (uint64_t)d.opt.sample_count;
	}
	for (const AobStats& s : all) stats->ms_trace = std::max(stats->ms_trace, s.ms_trace);
	return AOB_OK;
}

// C entry points of the two pure functions above, for the twin test against aobaker_amd/dist.py (tests/test_host_logic.py)
extern "C" int aobhost_rebalance_rows(const int32_t* bounds, const double* old_ms, int32_t parts, int32_t height, int32_t align, int32_t* out)
{
	if (!bounds || !old_ms || !out || parts < 1) return AOB_ERR_INVALID_ARGUMENT;
	const std::vector<int32_t> r = aob_rebalance_rows(std::vector<int32_t>(bounds, bounds + parts + 1), std::vector<double>(old_ms, old_ms + parts), height, align);
	std::copy(r.begin(), r.end(), out);
	return AOB_OK;
}

extern "C" int aobhost_equal_rows(int32_t height, int32_t parts, int32_t align, int32_t* out)
{
	if (!out || parts < 1) return AOB_ERR_INVALID_ARGUMENT;
	const std::vector<int32_t> r = aob_equal_rows(height, parts, align);
	std::copy(r.begin(), r.end(), out);
	return AOB_OK;
}
