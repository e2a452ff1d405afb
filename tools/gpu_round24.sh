#!/bin/bash
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -8
