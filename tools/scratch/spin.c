#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <stdint.h>
static volatile int stop;
static void *w(void *a){ uint64_t x=1, n=0; while(!stop){ for(int i=0;i<100000;i++) x=x*6364136223846793005ULL+1442695040888963407ULL; n++; } *(uint64_t*)a = n + (x&1); return 0; }
int main(int c,char**v){ int T=atoi(v[1]); pthread_t th[512]; uint64_t cnt[512]; for(int i=0;i<T;i++) pthread_create(&th[i],0,w,&cnt[i]); struct timespec ts={1,0}; nanosleep(&ts,0); stop=1; uint64_t s=0; for(int i=0;i<T;i++){pthread_join(th[i],0); s+=cnt[i];} printf("%d threads: %lu units/s, per thread %lu\n",T,(unsigned long)s,(unsigned long)(s/T)); }
