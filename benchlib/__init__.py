"""Support code of bench.py (the headline benchmark): committed-profile readers and rooflines (profiles), timing helpers (timing),
the traversal section's side measurements (traversal) and the renderer section (render).  bench.py holds the contract: arguments, the timed
region of the headline, the JSON line."""
