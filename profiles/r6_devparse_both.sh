bash profiles/r6_devparse_check.sh
bash profiles/r6_devparse_ab.sh
