#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <pthread.h>
#include <sched.h>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <atomic>
using namespace std::chrono;
static double now(){return duration<double,std::milli>(steady_clock::now().time_since_epoch()).count();}
volatile double sink;
void work(int us){ double t0=now(); double x=0; while(now()-t0 < us*1e-3) x+=1; sink=x; }
int main(int argc, char**argv){
  int us = atoi(argv[1]); int pin = atoi(argv[2]);
  cpu_set_t allowed; sched_getaffinity(0, sizeof(allowed), &allowed);
  std::vector<int> cpus; for(int c=0;c<CPU_SETSIZE;++c) if(CPU_ISSET(c,&allowed)) cpus.push_back(c);
  printf("allowed cpus: %zu\n", cpus.size());
  // spawn-per-use with pinning
  for(int rep=0;rep<3;++rep){
    double t0=now();
    std::vector<std::thread> th;
    for(int i=0;i<7;++i) th.emplace_back([=](){ if(pin){cpu_set_t s; CPU_ZERO(&s); CPU_SET(cpus[(i+1)%cpus.size()],&s); pthread_setaffinity_np(pthread_self(),sizeof(s),&s);} work(us);});
    work(us);
    for(auto&t:th) t.join();
    printf("spawn 7 (pin %d) x %d us: %.2f ms\n", pin, us, now()-t0);
  }
  // persistent pool
  struct Pool { std::mutex m; std::condition_variable cv, done; int gen=0, pending=0; bool quit=false; std::function<void(int)> job; std::vector<std::thread> th; } P;
  for(int i=0;i<7;++i) P.th.emplace_back([&,i](){ if(pin){cpu_set_t s; CPU_ZERO(&s); CPU_SET(cpus[(i+1)%cpus.size()],&s); pthread_setaffinity_np(pthread_self(),sizeof(s),&s);} int seen=0; for(;;){ std::unique_lock<std::mutex> l(P.m); P.cv.wait(l,[&]{return P.quit||P.gen!=seen;}); if(P.quit) return; seen=P.gen; auto j=P.job; l.unlock(); j(i); l.lock(); if(--P.pending==0) P.done.notify_one(); } });
  std::this_thread::sleep_for(milliseconds(50));
  for(int rep=0;rep<5;++rep){
    double t0=now();
    { std::lock_guard<std::mutex> l(P.m); P.job=[&](int){work(us);}; P.pending=7; ++P.gen; }
    P.cv.notify_all();
    work(us);
    { std::unique_lock<std::mutex> l(P.m); P.done.wait(l,[&]{return P.pending==0;}); }
    printf("pool 7 (pin %d) x %d us: %.2f ms\n", pin, us, now()-t0);
    std::this_thread::sleep_for(milliseconds(20));
  }
  { std::lock_guard<std::mutex> l(P.m); P.quit=true; } P.cv.notify_all(); for(auto&t:P.th) t.join();
}
