// fba_state.h -- layout of everything the engine keeps resident in HBM.
//
// Particle set of slot e, buffer b (two buffers; resampling gathers from one into the other,
// bufsel[e] says which one is live):
//     record : float  [2][E][N][Cs]   one particle = one contiguous record:
//                                       words [0, C)  Dirichlet counts (BAFlatModel order: phi, psi)
//                                       word  C       domain state index (int32 bits)
//                                       words (C, Cs) padding
//                                     Cs = next power of two >= C + 1 (min 4) while that is <= 64
//                                     words, else C + 1 rounded up to 4: a tiger particle (C = 24) is
//                                     exactly one 128-byte line, so a simulation's root sample, a
//                                     rejection attempt and a resample copy each touch ONE line.
//     weight : double [2][E][N]       (importance sampling only) SoA: the scans stream it coalesced
// Records are kept whole (AoS) because every consumer wants the whole particle: the search reads
// state + the rows it samples from, rejection/resampling move whole particles; and the per-slot
// source set (N * Cs * 4 B = 512 KB for tiger at N = 4096) stays L2-resident while its slot's
// workgroup gathers from it.
//
// Search tree of slot e: `max_nodes` fixed-size node records, slot-major, so one tree step
// touches one or two cache lines:
//     word 0            : ActionNode visit count
//     words 1..A        : ChanceNode visit counts
//     words cq_off..    : ChanceNode Q values (double[A], 8-B aligned)
//     words child_off.. : child node index per (action, observation), -1 = absent
#pragma once

#include <stdint.h>

#include "fba_device.h"

namespace fba {

struct DeviceState {
    // --- per-slot experiment position ---
    int32_t* run;       // [E] global run index
    int32_t* episode;   // [E]
    int32_t* t;         // [E] real time-step inside the episode (= History::length())
    uint8_t* active;    // [E] slot still has work
    uint8_t* need_update;  // [E] belief update pending (last env step was not terminal)
    uint8_t* need_reset;   // [E] resetDomainStateDistribution pending (new episode)
    uint8_t* need_init;    // [E] Belief::initiate pending (new run)
    uint8_t* adv;          // [E] set by env_kernel: 1 = next time-step, 2 = episode ended
    int32_t* env_state; // [E] true environment state
    double* ret;        // [E] discounted return so far
    double* disc;       // [E] accumulated discount
    int32_t* action;    // [E] last selected action
    int32_t* obs;       // [E] last real observation
    // --- belief ---
    uint8_t* bufsel;    // [E]
    double* p_weight;   // [2][E][N]
    float* p_rec;       // [2][E][N][Cs] particle records (counts | state | pad)
    float* p_rec_fc;    // same shape: the second filter of the reinvigoration (fully connected) / cheating (correct graph) belief
    uint8_t* bufsel_fc; // [E]
    double* lik;        // [E] cheating belief: CheatingReinvigoration::_likelihood; [E] holds the threshold
    uint8_t* lazy_reset;  // [E] rejection filter: resetDomainStateDistribution is pending -- particle i's state is the RESET-stream draw, not the record's word
    int32_t* fault;     // [1] 0, or 1 + the slot whose rejection update exceeded REJECT_MAX_ATTEMPTS (host reports it)
    uint8_t* cheat_pending;  // [E] the update of this tick pushed log(likelihood) below the threshold (mh-within-gibbs: an update has run, mh_kernel has work)
    // mh-within-gibbs belief: the run's (action, observation) history by episode; lik[e] is the log likelihood, lik[E] the threshold
    int16_t* mh_a;      // [E][episodes * horizon]
    int16_t* mh_o;
    int32_t* mh_ep_len; // [E][episodes + 1]
    int32_t* mh_n_ep;   // [E] episodes in the history, the open one included
    // history particles (C4): ONE record buffer per slot; a resample / reset gathers into a scratch pool shared by a chunk of
    // slots and is copied back (the second buffer was a quarter of a slot's memory, and slots are what C4 lacks)
    float* rec_scratch;       // (unused since the buffers are swapped, not copied)
    uint8_t* copy_pending;    // [E] the slot's new filter is in its scratch place's buffer: swap_buffers_kernel makes it the slot's
    int32_t single_rec;       // 1: p_rec holds E + scratch_slots buffers [N][Cs]; rec_buf says which is whose (records only; weights stay double-buffered)
    int32_t* rec_buf;         // [E + scratch_slots]: buffer of slot e; buffer of scratch place b at [E + b].  A resample / reset gathers into the
                              // place's buffer and the two indices are swapped -- no copy (the copy back was a quarter of C4's belief update)
    int32_t slot_base;        // first slot of the chunk a chunked launch works on
    int32_t scratch_slots;
    // budgeted searches: only a few slots need a belief update or a reset per launch, scattered over all of them -- the chunked launches
    // then run over a compacted list of those slots (block b of a launch works on slot slot_list[slot_base + b], scratch place b)
    int32_t* slot_list;       // [E] or null
    int32_t* list_count;      // [1]
    int32_t* list_count_host; // [1] pinned host memory: where the launcher reads the count (a pageable destination is staged by the runtime)
    // A/B switches of the launchers, read from the environment when the context is created (tests and same-box comparisons flip them per context):
    int8_t ab_rows_hbm;       // FBA_HIST_ROWS=hbm: history particles read the prior's rows from the padded tables (L2) instead of the deduplicated rows in LDS
    int8_t ab_hist_multi;     // FBA_HIST_MULTI=0 / 1: 1 = never, 2 = always update a history-particle slot by several workgroups (0: from 4 096 particles)
    int8_t ab_no_etiger;      // FBA_NO_ETIGER=1: the episodic tiger family on the general tree layout
    int8_t ab_lockstep;       // the trees of a search_hist2_kernel wave start their simulations together (FBA_HIST_LOCKSTEP=0: each on its own)
    int32_t* search_order;    // [E] (bucket-tree contexts): the slots sorted by the depth their searches have left, rewritten in front of every
                              // search launch -- tree k of the launch works on slot search_order[k], so a wave's trees run simulations of one length
    int32_t* scratch_idx;     // [E] the scratch place of a listed slot
    int32_t use_list;         // 1 in the DeviceState of a launch over the list
    // incubator belief (StructureIncubatorSampling.cpp): the weighted shadow filter, laid out like p_rec / p_weight / bufsel
    float* p_rec_sh;         // [2][E][N][Cs]
    double* p_weight_sh;     // [2][E][N]
    uint8_t* bufsel_sh;      // [E]
    uint8_t* need_update_sh; // [E] the shadow filter's copy of need_update (its importance update clears it)
    const int32_t* inc_order;  // [P.incub] WeightedFilter::leastLikely of uniform weights: which shadow particles are bred anew
    int32_t shadow;          // 1 in the DeviceState the shadow filter's kernels are launched with
    // nested belief (NestedBelief.cpp): per count particle a flat filter of P.nested domain states, double-buffered
    int32_t* nest_s;     // [2][E][N][P.nested]
    int32_t* nest_sel;   // [E] the buffer that holds the current filters
    double* nest_scan;   // [E][N] inclusive prefix sums (device order) of the count particles' weights
    double* nest_total;  // [E]
    float* mh_scratch;  // [E][mh_scratch_words]: three count blobs, T and O tables, messages, probabilities, the state sequence
    int32_t mh_scratch_words;
    int32_t* p_side;    // [E][N][side_w] importance filters: {new state, cells to increment} of the pending update (side_w = 1 + FS + FO;
                        // history particles: {new state, the step's entry})
    uint32_t* hist_cnt; // [E] history particles: how many entries of action a every record of the slot holds, 8 bits per action (fba_device.h)
    int32_t side_w;
    double* wscan;      // [E][N] scratch: inclusive device-order prefix sums of normalised weights
    double* ctot;       // [E][N/256 + 2] scratch: chunk totals / carries of the multi-workgroup filter
    double* is_tot;     // [E][2] total weight before normalisation, total of the normalised weights
    int32_t ctot_stride;
    int32_t is_multi;   // importance filter runs as several launches (N too large for one workgroup)
    const float* prior; // [Cs] prior record (state word unset); all zero increments when Problem::packed
    const float* prior_dense;  // packed particles: the prior count table itself
    const double* uni_scan; // [N] prefix sums of N uniform weights 1/N (device order)
    double uni_total;
    // --- tree ---
    int32_t* nodes;     // [E][max_nodes][node_words]
    int32_t max_nodes, node_words, cq_off, child_off, cn_off;   // cn_off: 1 = word 0 holds the visits, the action counts follow; 0 = no visits word (hashed trees, even A)
    // large observation spaces (A*O > 64): children live in a per-slot open-addressing table
    // of {code_lo, code_hi | epoch << 4, child, -} entries tagged with the slot's search epoch, so
    // the table is never cleared: an entry of an older search reads as empty
    int4* hash;         // [E][hmask + 1] or null (dense child table inside the node record)
    uint32_t hmask;
    int32_t hash_compact;  // 8-byte entries {epoch << 27 | code, child} instead (codes below 2^27): fba_kernels.hip child_get
    uint32_t* epoch;    // [E]
    const double* log1p_tab; // [sims + 1]
    // history-particle searches (search_hist2_kernel): the tree of a slot is ONE open-addressing table of 64-byte buckets, two to a 128-byte
    // line, keyed by (parent bucket, action, observation) | epoch << 28.  A bucket IS a node: {key, n0 | n1 << 16, n2 | n3 << 16, -},
    // {q0, q1}, {q2, q3}, and four 4-byte keys of nodes that exist but were never visited again (three quarters of a tree's nodes: they
    // have no statistics to keep).  The probe for a child returns its statistics in the same line: one trip to memory per tree level.
    uint4* bkt;           // [E][bkt_lines][8] or null (the node records + hash table above serve the search instead)
    int32_t bkt_lines;    // 128-byte lines per slot (buckets / 2)
    double* s_root;       // [E][6] budgeted searches: the parked root's {n0..n3} (as int32) and {q0..q3}
    // budgeted searches (Problem::search_budget > 0): a search that ran out of iterations is parked -- the root's statistics in its
    // node record, the rest here -- and resumed by the next launch
    int32_t* s_sim;       // [E] simulations done by the parked search (0 = none parked: the next launch starts a search)
    int32_t* s_nodes;     // [E] nodes created so far
    int32_t* s_depth;     // [E] deepest level reached so far
    uint8_t* search_done; // [E] or null: the slot's search finished in the last launch (env_kernel steps those slots only)
    // --- outputs ---
    double* returns;    // [runs][episodes]
    int32_t* lengths;   // [runs][episodes]
    int32_t runs_total; // run indices >= run_offset + runs_total are not executed (<0: unbounded)
    int32_t run_offset;
    unsigned long long* sim_steps;    // [E]
    unsigned long long* belief_steps; // [E]
    unsigned long long* env_steps;    // [E]
    double* ep_sums;    // [E][3] finished episodes of this slot: count, sum of returns, sum of squares
    unsigned long long* upd_particles; // [E] particles written by belief updates
    unsigned long long* upd_attempts;  // [E] rejection attempts / importance particles stepped
    unsigned long long* upd_entries;   // [E] history particles: sum over updates of N * (entries per record before the update)
    fba_trace_rec* cur;  // [E] record being assembled for the current tick
    fba_trace_rec* trace; // [trace_cap]
    uint32_t* trace_hist;  // [trace_cap][FBA_TRACE_HIST_BINS] or null: the filter's state histogram after the update of each record (cfg.trace = 2)
    int32_t* trace_count;
    int32_t trace_cap;
    int32_t trace_on;
};

}  // namespace fba
