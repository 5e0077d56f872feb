"""Cutting a rocprofv3 kernel trace into training steps.

A step of d_steps discriminator updates and one generator update holds d_steps + 1 optimiser launches and ends with
the last of them: the generator's adam_kernel also does the end-of-step bookkeeping (dpc_adam_step_slices_adv; until
round 3 a one-thread rng_advance_kernel behind it was the marker).  Steps are counted from the FRONT of the trace -
what bench.py issues after its timed steps (single launches timed again for the roofline figures) is no step."""


def step_ends(names, d_steps=1):
    """names: kernel names in start order -> index of the last kernel of every step."""
    per = int(d_steps) + 1
    adam = [i for i, n in enumerate(names) if n.split('(')[0].replace('void ', '').startswith('adam_kernel')]
    return [adam[k] for k in range(per - 1, len(adam), per)]
