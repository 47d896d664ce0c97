"""Known answers that pin the CPU oracle (and through it the HIP path).

A. Shipped with the reference (platforms/reference/tests/):
   v0.reference:4-7   Volume energy 1: 2287.78 / 2: -1415.27 / Energy: 872.514; after +2e-3 nm on atom 121, y:
                      Energy 872.576, Energy Change 0.0615433, Energy Change from Gradient 0.0619746 (lines 8-15)
   v1.reference:2-5   Energy: -2476.66; -2476.58; 0.0874992; 0.0886249
   input: platforms/reference/tests/gaussvol.dat (openmm_agbnp_plugin_amd/data/fixture264.dat), parameterisation of
   TestReferenceAGBNPForce.cpp:47-70, probe of TestReferenceAGBNPForce.cpp:117-127 (pmove=121, direction=1, 2e-3).
B. Recorded by the survey from its run of the unmodified reference sources,
   13 significant digits.  The survey's export of the .dms coordinates was rounded differently from
   openmm_agbnp_plugin_amd/data/*.dat (full repr), so the .dms systems agree to ~1e-8 relative, the .dat fixtures to 1e-13.
C. Tree statistics recorded by the survey.
"""

REFERENCE_PRINTED = {  # 6 significant digits, as printed by the reference's test program
    0: dict(e_vol1=2287.78, e_vol2=-1415.27, energy=872.514, energy_moved=872.576, change=0.0615433, change_from_gradient=0.0619746),
    1: dict(energy=-2476.66, energy_moved=-2476.58, change=0.0874992, change_from_gradient=0.0886249),
}
PROBE = dict(atom=121, direction=1, offset=2.0e-3)

SURVEY_ENERGIES = {  # name: (version 0, version 1, relative tolerance)
    "fixture264": (872.5144482745, -2476.6640250694, 2e-13),
    "fixture264_ocl": (940.4722445147, -1947.5053488966, 2e-13),
    "trpcage": (934.6174974604, -1965.5443412668, 2e-8),
    "1dwc": (12507.9229215959, -25557.9768109966, 2e-8),
    "2clr": (18007.3751625759, -30026.6479582717, 2e-8),
}

SURVEY_TREE = {  # name: (levels 2..7 node counts, max subtree under one atom, max children of a node)
    "fixture264": ([1804, 5277, 6021, 3425, 933, 73], 440, 33),
    "trpcage": ([1522, 3941, 4217, 2387, 624, 41], 326, 31),
    "1dwc": ([31630, 74366, 67112, 31922, 6788, 175], 376, 46),
    "2clr": ([48181, 115928, 103796, 48035, 9956, 288], 457, 43),
}
