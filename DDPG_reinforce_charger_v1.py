#!/usr/bin/env python3
"""Drop-in for `julia DDPG_reinforce_charger_v1.jl` (the reference's entry script): same environment variables (JOB_ID, TASK_ID,
GPU_ID), same data/ and out/ layout, the batched MI355X path underneath.  See <package>/main.py for the contract.

    JOB_ID=1179808 TASK_ID=1 GPU_ID=0 python DDPG_reinforce_charger_v1.py
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    importlib.import_module("master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd.main").main()
