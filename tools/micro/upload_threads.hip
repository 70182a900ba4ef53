// Dev microbenchmark: pageable host memory to the device, one hipMemcpyAsync against several threads with a stream each
// (build: hipcc --offload-arch=gfx950 -O3 -pthread -o tools/bin/upload_threads tools/micro/upload_threads.hip)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
	const size_t n_total = size_t(192) << 20; // bytes
	char *p_dev;
	if(hipMalloc((void**)&p_dev, n_total) != hipSuccess) return 1;
	for(int n_threads : {1, 2, 4, 8}) {
		for(int rep = 0; rep < 3; ++ rep) {
			std::vector<std::vector<char> > src(n_threads);
			const size_t n_piece = n_total / n_threads;
			for(int t = 0; t < n_threads; ++ t) {
				src[t].resize(n_piece);
				memset(src[t].data(), rep + t, n_piece); // fresh pageable memory every time, touched
			}
			const double t0 = now_ms();
			std::vector<std::thread> workers;
			for(int t = 0; t < n_threads; ++ t) {
				workers.emplace_back([&, t]() {
					hipSetDevice(0);
					hipStream_t s;
					hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
					hipMemcpyAsync(p_dev + t * n_piece, src[t].data(), n_piece, hipMemcpyHostToDevice, s);
					hipStreamSynchronize(s);
					hipStreamDestroy(s);
				});
			}
			for(auto &w : workers) w.join();
			const double t1 = now_ms();
			if(rep == 2)
				printf("%d thread(s): %.2f ms for %zu MB = %.1f GB/s\n", n_threads, t1 - t0, n_total >> 20, double(n_total) / (t1 - t0) * 1e-6);
		}
	}
	// through one pinned bounce buffer filled by 8 threads
	char *p_pin;
	if(hipHostMalloc((void**)&p_pin, n_total, hipHostMallocDefault) != hipSuccess) return 1;
	std::vector<char> src(n_total);
	memset(src.data(), 7, n_total);
	for(int rep = 0; rep < 3; ++ rep) {
		const double t0 = now_ms();
		const int n_threads = 8;
		const size_t n_chunk = size_t(16) << 20;
		hipStream_t s;
		hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
		for(size_t b = 0; b < n_total; b += n_chunk) {
			std::vector<std::thread> workers;
			const size_t n_piece = n_chunk / n_threads;
			for(int t = 0; t < n_threads; ++ t)
				workers.emplace_back([&, t, b]() { memcpy(p_pin + b + t * n_piece, src.data() + b + t * n_piece, n_piece); });
			for(auto &w : workers) w.join();
			hipMemcpyAsync(p_dev + b, p_pin + b, n_chunk, hipMemcpyHostToDevice, s);
		}
		hipStreamSynchronize(s);
		hipStreamDestroy(s);
		const double t1 = now_ms();
		if(rep == 2)
			printf("pinned bounce, 8 copy threads, 16 MB chunks: %.2f ms = %.1f GB/s\n", t1 - t0, double(n_total) / (t1 - t0) * 1e-6);
	}
	return 0;
}
