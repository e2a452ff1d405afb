#!/bin/bash
timeout 900 python -m pytest tests/ -x -q -m gpu -k "ilu or itsol or smoother or gmres or cg" 2>&1 | tail -12
