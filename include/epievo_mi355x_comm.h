/* epievo_mi355x_comm.h -- C ABI of the multi-GPU exchange layer of the MI355X epievo build:
 * RCCL (linked directly: librccl, xGMI between the GPUs of a node) behind four calls.
 *
 * The reference is a single-threaded CPU program: nothing in
 * /root/reference is replaced by this file.  It exists because the MCEM inner loop shards over
 * the sites of the genome: per EM iteration the shards exchange
 *   1. their edge columns with the two genome neighbours (halo refresh before reset(),
 *      /root/reference/src/prog/epievo_est_params_histories.cpp:236-241 is the loop it sits in), and
 *   2. the rows of the sufficient statistics J/D (one all-gather after run_mcmc(), :248).
 * Buffers are DEVICE memory on the rank's GPU (epv_pack_columns_dev / epv_blocks_to_rows of
 * include/epievo_mi355x.h write them); nothing is staged through the host.
 *
 * Two ways to make the ranks:
 *   epv_comm_init_all   every rank lives in THIS process (the C++ EM driver: one context and
 *                       one host thread per GPU) -> ncclCommInitAll;
 *   epv_comm_init_rank  one rank per process (torchrun-style launchers) -> ncclCommInitRank
 *                       with an id from epv_comm_get_unique_id passed around by the launcher.
 * RCCL refuses two ranks on one physical GPU.  When epv_comm_init_all is given a device list
 * with repeats (rehearsing an N-GPU run on a smaller box, EPV_DEVICES=0,0,0,0) the group runs
 * in LOOPBACK mode: the same calls in the same order, transfers done as device-to-device
 * copies at epv_comm_group_end.  epv_comm_is_rccl tells which one is active.
 *
 * Calling convention (both modes): bracket the per-rank calls of one exchange step with
 * epv_comm_group_start / epv_comm_group_end when one thread drives several ranks; then
 * epv_comm_sync every rank before touching the buffers.  All calls return 0 or an EPV_ERR_*
 * code of epievo_mi355x.h (EPV_ERR_HIP for HIP and RCCL failures); epv_comm_last_error has the text.
 */
#ifndef EPIEVO_MI355X_COMM_H
#define EPIEVO_MI355X_COMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct epv_comm epv_comm;
#define EPV_COMM_ID_BYTES 128

int epv_comm_init_all(int n_ranks, const int *devices, epv_comm **comms);
int epv_comm_get_unique_id(void *id);
int epv_comm_init_rank(int device, int world, int rank, const void *id, epv_comm **comm);
void epv_comm_destroy(epv_comm *comm);
const char *epv_comm_last_error(const epv_comm *comm);
int epv_comm_is_rccl(const epv_comm *comm);
int epv_comm_rank(const epv_comm *comm);
int epv_comm_world(const epv_comm *comm);

int epv_comm_group_start(void);
int epv_comm_group_end(void);

/* Halo exchange with the genome neighbours rank-1 ("prev") and rank+1 ("next"): send
 * bytes_prev bytes to prev and receive as many from it, likewise for next.  A side with
 * bytes == 0, and the outer sides of the first and last rank, are skipped. */
int epv_comm_exchange(epv_comm *comm, const void *d_send_prev, void *d_recv_prev, uint64_t bytes_prev,
                      const void *d_send_next, void *d_recv_next, uint64_t bytes_next);

/* d_recv[world][bytes] <- every rank's d_send[bytes], in rank (= genome) order */
int epv_comm_all_gather(epv_comm *comm, const void *d_send, void *d_recv, uint64_t bytes);

/* wait until this rank's part of the exchanges issued so far is complete */
int epv_comm_sync(epv_comm *comm);

#ifdef __cplusplus
}
#endif

#endif
