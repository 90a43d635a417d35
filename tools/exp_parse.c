/* parse-only harness: par_ingest with a no-op submit (tools/exp_parse.sh) */
#include <stdlib.h>
#define _GNU_SOURCE
#include "host.h"
#include <time.h>
#include <stdio.h>
static uint64_t nb=0,nr=0;
static void submit(void *arg, read_batch *b, int g){(void)arg;(void)g; nb+=b->nbases; nr+=b->nreads;}
void die(const char *fmt, ...){fprintf(stderr,"die %s\n",fmt);exit(1);}
int main(int argc,char**argv){ int nt=atoi(argv[2]); struct timespec a,b; clock_gettime(CLOCK_MONOTONIC,&a);
 int rc=par_ingest(argv[1],SEQ_FMT_FASTQ,nt,false,32u<<20,submit,NULL,NULL);
 clock_gettime(CLOCK_MONOTONIC,&b); double dt=(b.tv_sec-a.tv_sec)+(b.tv_nsec-a.tv_nsec)*1e-9;
 printf("rc %d threads %d reads %lu bases %lu  %.3f s  (%.2f GB/s of bases)\n",rc,nt,nr,nb,dt,nb/dt/1e9); return 0;}
