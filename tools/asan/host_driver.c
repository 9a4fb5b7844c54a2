/* Host-side exercise of the libevac C ABI for the AddressSanitizer build (tools/asan_host.sh): everything a
 * process can call without a GPU -- version / status strings, config validation, the failure paths of evac_create,
 * NULL-handle and NULL-argument conventions.  (GPU AddressSanitizer is not available on the pool; the device code is
 * covered by the parity tests.) */
#include <stdio.h>
#include <string.h>

#include "evac.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main(void) {
    CHECK(evac_version() == EVAC_VERSION);
    for (int s = 1; s >= -7; --s) CHECK(strlen(evac_status_string(s)) > 0);
    evac_config_t c;
    memset(&c, 0, sizeof c);
    c.number_of_pedestrians = 60; c.width = c.height = 1.0f; c.step_size = 0.01f; c.noise_coef = 0.2f; c.eps = 1e-8f;
    c.enslaving_degree = 1.0f; c.init_reward_each_step = -1.0f; c.max_timesteps = 2000;
    c.positions = EVAC_POS_GRAV; c.statuses = EVAC_STAT_NO; c.type = EVAC_TYPE_DICT; c.alpha = 3.0f;
    CHECK(evac_config_validate(&c) == EVAC_OK && evac_config_obs_dim(&c) == 6);
    for (int n = -3; n <= 1030; n += 7) {                      /* sweep the pedestrians bound */
        c.number_of_pedestrians = n;
        const int rc = evac_config_validate(&c);
        CHECK((rc == EVAC_OK) == (n >= 1 && n <= EVAC_MAX_PEDESTRIANS));
        if (rc != EVAC_OK) CHECK(strlen(evac_last_error(NULL)) > 0 && evac_config_obs_dim(&c) == -1);
    }
    c.number_of_pedestrians = 60;
    for (int pos = -1; pos <= 3; ++pos) for (int st = -1; st <= 3; ++st) for (int ty = -1; ty <= 2; ++ty) {
        c.positions = pos; c.statuses = st; c.type = ty;
        const int rc = evac_config_validate(&c);
        const long long d = evac_config_obs_dim(&c);
        CHECK((rc == EVAC_OK) == (d > 0));
    }
    c.positions = EVAC_POS_REL; c.statuses = EVAC_STAT_OHE; c.type = EVAC_TYPE_BOX;
    CHECK(evac_config_obs_dim(&c) == 62 * 6);
    CHECK(evac_config_validate(NULL) == EVAC_ERR_INVALID_ARGUMENT);
    evac_handle_t h = (evac_handle_t)0x1;
    CHECK(evac_create(&c, 4, 0, 0, 0, NULL) == EVAC_ERR_INVALID_ARGUMENT);
    CHECK(evac_create(NULL, 4, 0, 0, 0, &h) == EVAC_ERR_INVALID_ARGUMENT && h == NULL);
    CHECK(evac_create(&c, 0, 0, 0, 0, &h) == EVAC_ERR_INVALID_ARGUMENT && h == NULL);
    const int rc = evac_create(&c, 4, 0, 7, 0, &h);            /* no GPU here: a loud failure, not a CPU fallback */
    if (rc == EVAC_OK) {                                       /* (on a GPU box the handle is simply created and destroyed) */
        CHECK(h != NULL && evac_obs_dim(h) == 62 * 6 && evac_num_envs(h) == 4);
        CHECK(strlen(evac_kernel_variant(h, 1)) > 0);
        CHECK(evac_step(h, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL) == EVAC_ERR_NOT_BOUND);
        CHECK(evac_destroy(h) == EVAC_OK);
    } else {
        CHECK(rc == EVAC_ERR_NO_DEVICE && h == NULL && strstr(evac_last_error(NULL), "no CPU path") != NULL);
    }
    /* NULL handles never crash */
    CHECK(evac_obs_dim(NULL) == -1 && evac_num_envs(NULL) == -1 && evac_algorithmic_bytes_per_env_step(NULL) == -1);
    CHECK(evac_step(NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL) == EVAC_ERR_INVALID_ARGUMENT);
    CHECK(evac_rollout(NULL, 1, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL) == EVAC_ERR_INVALID_ARGUMENT);
    CHECK(evac_reset(NULL, NULL, NULL, NULL, NULL) == EVAC_ERR_INVALID_ARGUMENT);
    CHECK(evac_observe(NULL, NULL, NULL) == EVAC_ERR_INVALID_ARGUMENT);
    CHECK(evac_bind_state(NULL, NULL, NULL, NULL, NULL, NULL) == EVAC_ERR_INVALID_ARGUMENT);
    CHECK(evac_norm_init(NULL, NULL, NULL) == EVAC_ERR_INVALID_ARGUMENT && evac_norm_state_doubles(NULL) == -1);
    CHECK(strlen(evac_kernel_variant(NULL, 0)) == 0 && evac_destroy(NULL) == EVAC_OK);
    printf("asan host driver: ok\n");
    return 0;
}
