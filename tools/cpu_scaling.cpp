#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstring>
int main(){ for (int T : {1,2,4,8}) { auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> th; std::vector<double> out(T);
 for(int i=0;i<T;++i) th.emplace_back([&,i]{ double a=0; for (long k=0;k<200000000L/ T;++k) a+= (double)(k^i)*1e-9; out[i]=a;}); for(auto&t:th)t.join();
 printf("T=%d %.3f s\n",T,std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count()); }
 // memcpy bandwidth
 std::vector<char> a(1<<28,1), b(1<<28);
 for (int T : {1,4,8}) { auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> th; size_t n=a.size();
 for(int i=0;i<T;++i) th.emplace_back([&,i]{ memcpy(b.data()+n*i/T, a.data()+n*i/T, n/T);}); for(auto&t:th)t.join();
 printf("memcpy 256MB T=%d %.3f s\n",T,std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count()); }
}
