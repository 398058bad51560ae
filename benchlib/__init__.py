"""bench.py's parts: common (request, loading, roofline constants), headline (the timed region), extras, workloads (the other BASELINE
configs), cpu_baseline.  bench.py is the orchestrator and re-exports all of it (scripts/ import it as one module)."""
