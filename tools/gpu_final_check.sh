#!/bin/bash
# The round's closing check on the final build: whole GPU suite (sanitizer tests must RUN: GMSM_REQUIRE_SANITIZERS=1 turns
# their skip into a failure, -rs prints any other skip), smoke(), the bench line.  = tools/gpu_session.sh <label> suite smoke bench
exec "$(dirname "$0")/gpu_session.sh" "${1:-final}" suite smoke bench
