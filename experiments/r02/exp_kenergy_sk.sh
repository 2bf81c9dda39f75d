#!/bin/bash
cd /root/repo
for rep in 1 2; do
  ./tools/kenergy ffn2 0 3
  KENERGY_SPLITK=1 ./tools/kenergy ffn2 0 3
done
