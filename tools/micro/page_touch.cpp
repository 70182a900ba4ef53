// dev helper (round 6): what first touches of fresh anonymous memory cost on the box, by who touches and how the pages come:
// one thread / eight threads on 4 KB pages, MAP_POPULATE, transparent huge pages (madvise) -- the floor under the host analysis
// g++ -O2 -pthread -o page_touch page_touch.cpp
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void touch(char *p, size_t n_begin, size_t n_end, size_t n_step)
{
	for(size_t i = n_begin; i < n_end; i += n_step)
		p[i] = 1;
}

static void run(const char *p_s_what, size_t n_bytes, int n_threads, int n_map_flags, bool b_huge, bool b_fill)
{
	const double t0 = now();
	char *p = (char*)mmap(0, n_bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | n_map_flags, -1, 0);
	if(p == (char*)MAP_FAILED) {
		printf("%-58s mmap failed\n", p_s_what);
		return;
	}
	char *q = (char*)((size_t(p) + (2 << 20) - 1) / (2 << 20) * (2 << 20));
	if(b_huge)
		(void)madvise(q, n_bytes, MADV_HUGEPAGE);
	const double t1 = now();
	std::vector<std::thread> threads;
	for(int t = 0; t < n_threads; ++ t) {
		const size_t n_begin = n_bytes / n_threads * t, n_end = n_bytes / n_threads * (t + 1);
		if(b_fill)
			threads.emplace_back([=]() { memset(q + n_begin, 1, n_end - n_begin); });
		else
			threads.emplace_back(touch, q, n_begin, n_end, size_t(4096));
	}
	for(size_t t = 0; t < threads.size(); ++ t)
		threads[t].join();
	const double t2 = now();
	(void)munmap(p, n_bytes + (2 << 20));
	const double t3 = now();
	printf("%-58s map %7.2f ms, %s %7.2f ms (%5.1f GB/s), unmap %6.2f ms\n", p_s_what, t1 - t0, b_fill? "fill " : "touch", t2 - t1,
		n_bytes / (t2 - t1) * 1e-6, t3 - t2);
}

int main()
{
	const size_t n = size_t(256) << 20;
	for(int n_round = 0; n_round < 2; ++ n_round) {
		run("256 MB, 4 KB pages, touched by 1 thread", n, 1, 0, false, false);
		run("256 MB, 4 KB pages, touched by 4 threads", n, 4, 0, false, false);
		run("256 MB, 4 KB pages, touched by 8 threads", n, 8, 0, false, false);
		run("256 MB, 4 KB pages, touched by 16 threads", n, 16, 0, false, false);
		run("256 MB, 4 KB pages, filled by 1 thread", n, 1, 0, false, true);
		run("256 MB, 4 KB pages, filled by 8 threads", n, 8, 0, false, true);
		run("256 MB, MAP_POPULATE, then touched by 1 thread", n, 1, MAP_POPULATE, false, false);
		run("256 MB, huge pages asked for, touched by 1 thread", n, 1, 0, true, false);
		run("256 MB, huge pages asked for, touched by 8 threads", n, 8, 0, true, false);
		run("256 MB, huge pages asked for, filled by 8 threads", n, 8, 0, true, true);
	}
	return 0;
}
