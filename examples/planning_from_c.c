/* examples/planning_from_c.c -- the C-ABI from plain C99: experiment::planning::run (PlanningExperiment.cpp:27-55) on the
 * episodic tiger POMDP, all runs concurrently on the GPU.
 *   gcc -std=c99 -Iinclude examples/planning_from_c.c -Lfba_pomdp_amd -lfba_hip -Wl,-rpath,$PWD/fba_pomdp_amd -lm -o planning_from_c
 *   ./planning_from_c [runs] [simulations] [particles]                                                                       */
#include <stdio.h>
#include <stdlib.h>

#include "fba_hip.h"

int main(int argc, char** argv)
{
    fba_config cfg;
    fba_ctx* ctx = NULL;
    fba_stat st;
    fba_default_config(&cfg);
    cfg.domain    = FBA_DOM_TIGER_EPISODIC;
    cfg.model     = FBA_MODEL_POMDP;
    cfg.belief    = FBA_BELIEF_REJECTION;
    cfg.runs      = argc > 1 ? atoi(argv[1]) : 1000;
    cfg.sims      = argc > 2 ? atoi(argv[2]) : 1024;
    cfg.particles = argc > 3 ? atoi(argv[3]) : 256;
    cfg.seed      = 7;
    if (fba_create(&cfg, &ctx) != FBA_OK) {
        fprintf(stderr, "fba_create: %s\n", fba_last_error(NULL));
        return 1;
    }
    if (fba_run_planning(ctx, &st) != FBA_OK) {
        fprintf(stderr, "fba_run_planning: %s\n", fba_last_error(ctx));
        fba_destroy(ctx);
        return 1;
    }
    printf("runs %g  mean return %.6g  stder %.6g\n", st.count, st.mean, fba_stat_stder(&st));
    fba_destroy(ctx);
    return 0;
}
