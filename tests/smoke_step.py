"""One tiny D+G training step on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke())."""


def run():
    import test_engine_gpu as t
    t.test_engine_sampler_path_matches_oracle_on_same_draws()
    print('smoke ok: one D step + one G step on cuda:0 match the CPU oracle')
