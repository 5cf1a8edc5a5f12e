"""Starting the Fortran programs of this repo (the compiled reference, its drop-in builds) as child processes.

One helper for tests, fixture scripts and bench.py: no shell hop, no `exec` in a process that may carry a profiler's preload.
The reference's automatic arrays (idum(0:kk), thpsi(18,kk), ...) overflow the default 8 MB stack, so the child gets an unlimited
stack through setrlimit in its own pre-exec hook, and an environment from which every profiler preload has been
removed (a child that inherits LD_PRELOAD of rocprofv3 would initialise the GPU before main())."""
import os
import resource
import subprocess

_PROFILER_PREFIXES = ("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTX", "ROCTRACER")


def scrubbed_env(extra=None, base=None):
    """Copy of the environment without LD_PRELOAD and without the profiler's variables, updated with `extra`."""
    env = {k: v for k, v in (os.environ if base is None else base).items() if k != "LD_PRELOAD" and not k.startswith(_PROFILER_PREFIXES)}
    if extra:
        env.update({k: str(v) for k, v in extra.items()})
    return env


def under_profiler():
    return any(k.startswith(("ROCP", "ROCPROF")) or k == "HSA_TOOLS_LIB" for k in os.environ)


def _unlimited_stack():
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


def run_with_unlimited_stack(argv, cwd=None, env=None, timeout=None, omp_stacksize="1G", scrub=True, **kw):
    """subprocess.run(argv) with RLIMIT_STACK = unlimited in the child and OMP_STACKSIZE set; captures text output.
    `env`: extra variables (merged into the scrubbed environment), not a whole environment."""
    e = scrubbed_env(env) if scrub else dict(os.environ, **{k: str(v) for k, v in (env or {}).items()})
    e.setdefault("OMP_STACKSIZE", omp_stacksize)
    if isinstance(argv, str):
        argv = [argv]
    return subprocess.run(list(argv), cwd=cwd, env=e, capture_output=True, text=True, timeout=timeout, preexec_fn=_unlimited_stack, **kw)
