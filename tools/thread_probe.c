#include <pthread.h>
#include <stdio.h>
#include <time.h>
static double now(){struct timespec ts;clock_gettime(CLOCK_MONOTONIC,&ts);return ts.tv_sec*1e3+ts.tv_nsec*1e-6;}
static void* w(void*a){volatile double x=0;for(long i=0;i<200000000;i++)x+=i;return 0;}
int main(){for(int n=1;n<=8;n*=2){pthread_t t[8];double t0=now();for(int i=0;i<n;i++)pthread_create(&t[i],0,w,0);for(int i=0;i<n;i++)pthread_join(t[i],0);printf("%d threads: %.0f ms\n",n,now()-t0);}}
