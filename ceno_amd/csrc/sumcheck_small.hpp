// The latency ladder of a pipelined generic sumcheck (sumcheck_small.hip): term-parallel tiles (k_tile), persistent mid rounds
// over resident workgroups (k_mid) and the persistent single-workgroup tail (k_tail).  sumcheck.hip decides which rounds go
// where with the eligibility / geometry helpers below and launches through these entry points.
#pragma once
#include "sumcheck_dev.hpp"

// one 64-byte relay line per workgroup of a k_mid launch (used when the mailbox lives in host memory)
struct alignas(64) MidRelay {
    unsigned long long seq;   // (nonce << 8) | round + 1, or (nonce << 8) | 0xFF: give up
    unsigned long long pad;
    unsigned long long chal[2];
    unsigned long long pad2[4];
};

int tile_pairs(size_t n_flat, size_t n_mles, size_t pairs);
bool tile_eligible(size_t n_mles, size_t pairs);
bool tail_eligible(size_t n_mles, size_t pairs, int d, size_t n_flat);
void mid_geometry(size_t n_mles, size_t pairs, int d, size_t n_flat, int w_cap, int* W, int* S0);
// resident k_mid<d> workgroups per compute unit at the dynamic LDS of slices of S0 pairs (runtime occupancy query, cached); 0 = none fit
int mid_blocks_per_cu(int d, size_t n_mles, size_t S0, size_t n_flat);
void launch_tile(int d, const DevPlan& pl, int n_mles, int n_flat, size_t pairs, E2 r, const Epilogue& ep, hipStream_t st);
// rounds i0 .. n-1 on the device; export_host != NULL: the tables of round n-1 (2 * pairs entries each) follow its message into
// pinned host memory and the host runs the rounds from n on (sumcheck.hip: sc_host_round)
void launch_tail(int d, const DevPlan& pl, const MleSlot* last_slots, int n_mles, int n_flat, size_t pairs, int i0, int n, const Epilogue& ep,
                 E2* out_evals, E2* export_host, hipStream_t st);
void launch_mid(int d, const DevPlan& pl, const MleSlot* out_slots, int n_mles, int n_flat, int W, int S0, int i0, int i1, const Epilogue& ep,
                MidRelay* relay, unsigned long long nonce, int direct_poll, hipStream_t st);
